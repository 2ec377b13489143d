"""CPU: magnet_amd.fnet.PSMNet (the torch path of the F-Net) against the golden samples captured from the
reference's PSMNet (G10), state_dict compatibility, and the BatchNorm folding / weight re-packing used by FNetMFMA."""
import numpy as np
import torch

from magnet_amd import fnet
from tests.stubs import procedural_images, seeded_fnet_state


def test_G10_psmnet_matches_reference(golden):
    m = seeded_fnet_state(fnet.PSMNet(feature_dim=64), seed=10).eval()
    with torch.no_grad():
        out = m(procedural_images(2, 256, 320))
    assert tuple(out.shape) == (2, 64, 64, 80)
    scale = float(np.abs(golden["G10_feat_sparse"]).max())
    np.testing.assert_allclose(out[:, :, ::4, ::4].numpy(), golden["G10_feat_sparse"], rtol=0, atol=2e-5 * scale)
    np.testing.assert_allclose(out.abs().mean(dim=(0, 2, 3)).numpy(), golden["G10_feat_absmean"], rtol=1e-4)


def test_state_dict_layout():
    sd = fnet.PSMNet(feature_dim=64).state_dict()
    assert len(sd) == 361 and sum(v.numel() for k, v in sd.items() if not k.endswith("num_batches_tracked") and "running" not in k) == 3343648
    for k in ("firstconv.0.0.weight", "firstconv.4.1.running_var", "layer1.2.conv1.0.0.weight", "layer2.0.downsample.1.bias",
              "layer2.15.conv2.1.weight", "layer4.2.conv2.0.weight", "branch3.1.0.weight", "lastconv.0.1.running_mean", "lastconv.2.weight"):
        assert k in sd, k
    assert sd["layer2.0.conv1.0.0.weight"].shape == (64, 32, 3, 3) and sd["layer3.0.downsample.0.weight"].shape == (128, 64, 1, 1)

    class A:
        FNET_architecture = "PSM-Net"; FNET_feature_dim = 64
    assert all(k.startswith("f_net.") for k in fnet.FNET(A()).state_dict())


def test_bn_fold_and_s2d_weights():
    """conv+BN(eval) == folded conv; stride-2 3x3 == 2x2-window conv over the space-to-depth tensor with _pack_s2d weights."""
    torch.manual_seed(0)
    seq = fnet._conv_bn(8, 16, 3, stride=2).eval()
    seq[1].running_mean.normal_(); seq[1].running_var.uniform_(0.5, 2.0); seq[1].weight.data.uniform_(0.5, 1.5); seq[1].bias.data.normal_()
    x = torch.randn(2, 8, 10, 14)
    w, b = fnet._fold(seq)
    ref = seq(x)
    np.testing.assert_allclose(torch.nn.functional.conv2d(x, w, b, stride=2, padding=1).detach().numpy(), ref.detach().numpy(), atol=2e-5)
    hi, lo = fnet._pack_s2d(w)                                   # (4, 16, 32) planes
    w4 = (hi.float() + lo.float())                               # bf16x2 reconstruction (16 mantissa bits)
    s2d = torch.cat([x[:, :, py::2, px::2] for py in (0, 1) for px in (0, 1)], dim=1)          # (2, 32, 5, 7)
    s2d = torch.nn.functional.pad(s2d, (1, 0, 1, 0))             # taps reach (-1,-1)
    k = w4.reshape(2, 2, 16, 32).permute(2, 3, 0, 1)             # (cout, cin, ty, tx)
    got = torch.nn.functional.conv2d(s2d, k, b)
    np.testing.assert_allclose(got.detach().numpy(), ref.detach().numpy(), atol=2e-3)            # bf16x2 weight rounding only
