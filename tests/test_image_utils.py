"""mse / psnr of the reference's utils/image_utils.py:17-22 (the file itself needs jax and is not importable here: the two
expressions are restated in the test with torch on the CPU, which is what the reference evaluates)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref_mse(a, b):  # utils/image_utils.py:17-18
    return (((a - b)) ** 2).view(a.shape[0], -1).mean(1, keepdim=True)


def _ref_psnr(a, b):  # utils/image_utils.py:20-22
    return 20 * torch.log10(1.0 / torch.sqrt(_ref_mse(a, b)))


@pytest.mark.parametrize("shape", [(3, 1080, 1920), (1, 3, 270, 480), (3, 7, 5), (5, 1), (2, 3, 100, 333)])
def test_mse_and_psnr_match_the_reference_expressions(shape):
    from based_on_point_cloud_partitions_amd.image_utils import mse, psnr

    g = torch.Generator().manual_seed(sum(shape))
    a = torch.rand(shape, generator=g)
    b = (a + 0.05 * torch.randn(shape, generator=g)).clamp(0, 1)
    got_m, got_p = mse(a.cuda(), b.cuda()).cpu(), psnr(a.cuda(), b.cuda()).cpu()
    want_m, want_p = _ref_mse(a.double(), b.double()), _ref_psnr(a.double(), b.double())
    assert got_m.shape == (shape[0], 1) and got_p.shape == (shape[0], 1)
    assert torch.allclose(got_m.double(), want_m, rtol=2e-6, atol=0) and torch.allclose(got_p.double(), want_p, rtol=0, atol=2e-5)
    assert torch.equal(mse(a.cuda(), b.cuda()).cpu(), got_m)  # fixed summation order


def test_mse_rejects_cpu_tensors_and_psnr_of_identical_images_is_inf():
    from based_on_point_cloud_partitions_amd import GsrError
    from based_on_point_cloud_partitions_amd.image_utils import mse, psnr

    a = torch.rand(3, 8, 8)
    with pytest.raises(GsrError):
        mse(a, a)
    assert torch.isinf(psnr(a.cuda(), a.cuda())).all()
