"""GPU parity of the harness-side ops of inference.py vs oracle/harness_ref.py."""
import numpy as np
import pytest
import torch

from tests.gpu_util import DEV
from tests.helpers import bits_equal
from tests.synth import rect_mask, synth_input

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(2, 3, 16, 16), (3, 13, 9, 7), (1, 3, 256, 256)])
def test_cond_assembly_postprocess_preview_bit_exact(shape):
    from eo_diffusion_amd import harness as H
    from oracle import harness_ref as R
    n, c, h, w = shape
    image = synth_input("h_img", shape, 1, uniform=True)
    seg = 1.0 - rect_mask(n, h, w, 2)  # the dataset's segmentation: 1 = region to repaint
    cond = H.assemble_repaint_cond(image.to(DEV), seg.to(DEV)).cpu()
    assert bits_equal(cond, R.repaint_cond(image, seg))
    assert cond.shape == (n, c + 1, h, w)
    samples = synth_input("h_smp", shape, 3, scale=0.8)
    for img in (image, image * 2 - 1):  # [0,1] data -> clip, [-1,1] data -> (x+1)/2
        got = H.postprocess_samples(samples.to(DEV), img.to(DEV)).cpu()
        assert bits_equal(got, R.postprocess(samples, img))
    assert bits_equal(H.postprocess_samples(samples.to(DEV), data_nonneg=True).cpu(), samples.clip(0, 1))
    keep = 1.0 - seg
    assert bits_equal(H.masked_preview(image.to(DEV), keep.to(DEV)).cpu(), R.masked_preview(image, keep))


def test_psnr_ssim_vs_published_definitions():
    from eo_diffusion_amd import harness as H
    from oracle import harness_ref as R
    gt = synth_input("m_gt", (2, 3, 40, 48), 1, uniform=True)
    pred = (gt + 0.05 * synth_input("m_nz", (2, 3, 40, 48), 2)).clip(0, 1)
    assert abs(H.psnr(pred.to(DEV), gt.to(DEV)) - R.psnr(pred, gt)) < 1e-4
    assert abs(H.ssim(pred.to(DEV), gt.to(DEV)) - R.ssim(pred, gt)) < 1e-9
    assert H.ssim(gt, gt) == pytest.approx(1.0, abs=1e-12) and H.psnr(gt.to(DEV), gt.to(DEV)) == float("inf")
