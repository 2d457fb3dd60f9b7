"""GPU parity: fused sampler kernels vs the CPU oracle and the golden vectors -- BIT-EXACT."""
import numpy as np
import pytest
import torch

from oracle import sampler_ref as SR
from oracle import schedule as SCH
from tests.helpers import bits_equal, close_ulp, gt
from tests.synth import synth_input

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _model(T, size=8, cond_type=None):
    from eo_diffusion_amd.diffusion.model import EODiffusion

    class Fixed(torch.nn.Module):
        def forward(self, x, t, cond=None, y=None):
            return self.pred

    return EODiffusion(Fixed(), timesteps=T, image_size=size, in_channels=3, cond_type=cond_type).to(DEV)


def test_schedule_buffers():
    for T in (20, 200, 1000):
        m = _model(T)
        g = gt(f"schedule_T{T}")
        for k, v in g.items():
            if k.startswith("sqrt_"):  # host torch.sqrt is CPU-model dependent (<= 1 ulp)
                assert close_ulp(getattr(m, k).cpu(), v), k
            else:
                assert bits_equal(getattr(m, k).cpu(), v), k


def _with_tables(m, tb):
    """give the product model exactly the oracle's tables so kernel-vs-oracle can be compared bit for bit"""
    for k, v in tb.items():
        getattr(m, k).copy_(v)
    return m


@pytest.mark.parametrize("tag", ["t999", "t500", "t1", "t0", "tmix0", "tmix"])
def test_steps_vs_golden(tag):
    g = gt("sampler_steps_T1000")
    m = _model(1000)
    t, x, pred, noise = (g[tag + s].to(DEV) for s in ("_t", "_x", "_pred", "_noise"))
    m.model.pred = pred
    # vs the reference's CPU output (its sqrt is off by <= 1 ulp, CPU dependent) ...
    assert close_ulp(m._reverse_diffusion_with_clip(x, t, noise).cpu(), g[tag + "_clip"])
    assert close_ulp(m._reverse_diffusion(x, t, noise).cpu(), g[tag + "_noclip"])
    assert close_ulp(m._forward_diffusion(x, t, noise).cpu(), g[tag + "_q"])
    # ... and BIT-EXACT vs the oracle (same op sequence, IEEE sqrt) on identical tables
    tb = SCH.eo_cosine_tables(1000)
    _with_tables(m, tb)
    tc, xc, pc, nc = t.cpu(), x.cpu(), pred.cpu(), noise.cpu()
    assert bits_equal(m._reverse_diffusion_with_clip(x, t, noise).cpu(), SR.ddpm_step_clip(tb, xc, tc, nc, pc))
    assert bits_equal(m._reverse_diffusion(x, t, noise).cpu(), SR.ddpm_step_noclip(tb, xc, tc, nc, pc))
    assert bits_equal(m._forward_diffusion(x, t, noise).cpu(), SR.q_sample(tb, xc, tc, nc))


def test_steps_sweep_all_t_vs_oracle():
    """every timestep of T=1000, ragged tensor size (not a multiple of 4 or 256)"""
    tb = SCH.eo_cosine_tables(1000)
    m = _with_tables(_model(1000), tb)
    x = synth_input("swx", (1, 3, 5, 7), 21)
    pred = synth_input("swp", (1, 3, 5, 7), 21)
    noise = synth_input("swn", (1, 3, 5, 7), 21)
    for ti in list(range(0, 1000, 37)) + [1, 2, 3, 998, 999]:
        t = torch.tensor([ti])
        m.model.pred = pred.to(DEV)
        a = m._reverse_diffusion_with_clip(x.to(DEV), t.to(DEV), noise.to(DEV)).cpu()
        assert bits_equal(a, SR.ddpm_step_clip(tb, x, t, noise, pred)), ti
        b = m._reverse_diffusion(x.to(DEV), t.to(DEV), noise.to(DEV)).cpu()
        assert bits_equal(b, SR.ddpm_step_noclip(tb, x, t, noise, pred)), ti


def test_repaint_mix_bit_exact():
    tb = SCH.eo_cosine_tables(1000)
    m = _with_tables(_model(1000, size=16, cond_type="sum"), tb)
    from tests.synth import rect_mask
    x = synth_input("rx", (3, 3, 16, 16), 22)
    g0 = synth_input("rg", (3, 3, 16, 16), 22, uniform=True)
    nz = synth_input("rn", (3, 3, 16, 16), 22)
    mask = rect_mask(3, 16, 16, 22)
    t = torch.tensor([0, 400, 999])
    out = m._repaint_mix(x.to(DEV), g0.to(DEV), mask.to(DEV), t.to(DEV), nz.to(DEV)).cpu()
    assert bits_equal(out, SR.repaint_mix(tb, x, g0, mask, t, nz))


def test_repaint_mix_with_a_mask_per_channel_bit_exact():
    """`img_orig * mask + (1. - mask) * img` (ddim.py:147-148) broadcasts: a mask that differs between channels is legal upstream"""
    tb = SCH.eo_cosine_tables(1000)
    m = _with_tables(_model(1000, size=16, cond_type="sum"), tb)
    x = synth_input("rx", (3, 3, 16, 16), 22)
    g0 = synth_input("rg", (3, 3, 16, 16), 22, uniform=True)
    nz = synth_input("rn", (3, 3, 16, 16), 22)
    mask = (synth_input("rm", (3, 3, 16, 16), 23) > 0.3).float()
    t = torch.tensor([0, 400, 999])
    out = m._repaint_mix(x.to(DEV), g0.to(DEV), mask.to(DEV), t.to(DEV), nz.to(DEV)).cpu()
    assert bits_equal(out, SR.repaint_mix(tb, x, g0, mask, t, nz))


def test_ddim_mask_shapes_broadcast_like_the_reference():
    """DDIMSampler.ddim_sampling takes every mask the reference's broadcast takes -- [H,W] (also a numpy array: make_label's output),
    [1,1,H,W], [N,1,H,W], [N,C,H,W] -- with bit-identical results where they describe the same mask; a shape that does not broadcast is
    an EodError naming it"""
    from eo_diffusion_amd._lib import EodError
    from eo_diffusion_amd.diffusion.ddim import DDIMSampler
    from tests.synth import rect_mask
    m = _model(1000, size=16)
    m.model.pred = synth_input("dp", (2, 3, 16, 16), 5).to(DEV)
    s = DDIMSampler(m)
    s.make_schedule(ddim_num_steps=5, ddim_eta=0.5, verbose=False)
    xT, x0 = synth_input("dx", (2, 3, 16, 16), 6), synth_input("d0", (2, 3, 16, 16), 7, uniform=True)
    sn = [synth_input(f"ds{i}", (2, 3, 16, 16), 8 + i) for i in range(5)]
    mn = [synth_input(f"dm{i}", (2, 3, 16, 16), 18 + i) for i in range(5)]
    one = rect_mask(1, 16, 16, 3)  # [1,1,16,16]

    def run(mask, x0_=x0):
        out, _ = s.ddim_sampling(None, (2, 3, 16, 16), x_T=xT, mask=mask, x0=x0_, step_noises=sn, mix_noises=mn, progress=False)
        return out.cpu()

    ref = run(one.expand(2, 1, 16, 16).contiguous())
    for mk in (one, one[0, 0], one[0, 0].numpy(), one[0], one.expand(2, 3, 16, 16), one.to(DEV)):
        assert bits_equal(run(mk), ref)
    assert bits_equal(run(one, x0.numpy()), ref)  # (x0 from the host: moved once, the mix noise is drawn on the device)
    assert not bits_equal(run(1.0 - one), ref)
    with pytest.raises(EodError, match="broadcast"):
        run(torch.ones(2, 1, 8, 8))
    with pytest.raises(EodError, match="broadcast"):
        run(torch.ones(3, 1, 16, 16))


@pytest.mark.parametrize("eta", [0.0, 0.7])
def test_ddim_steps_vs_golden(eta):
    from eo_diffusion_amd.diffusion.ddim import DDIMSampler
    g = gt("ddim_steps_S250_T1000")
    m = _model(1000)
    s = DDIMSampler(m)
    s.make_schedule(ddim_num_steps=250, ddim_eta=eta, verbose=False)
    assert np.array_equal(np.asarray(s.ddim_timesteps), SCH.ddim_timesteps("uniform", 250, 1000))
    for index in (249, 100, 1, 0):
        k = f"eta{eta}_i{index}_"
        m.model.pred = g[k + "e"].to(DEV)
        t = torch.full((2,), int(s.ddim_timesteps[index]), dtype=torch.long, device=DEV)
        xp, p0 = s.p_sample_ddim(g[k + "x"].to(DEV), None, t, index=index, _noise=g[k + "noise"])
        assert close_ulp(xp.cpu(), g[k + "x_prev"]) and close_ulp(p0.cpu(), g[k + "pred_x0"])
        oxp, op0 = SR.ddim_step(g[k + "x"], g[k + "e"], s.ddim_alphas[index], s.ddim_alphas_prev[index], s.ddim_sigmas[index],
                                s.ddim_sqrt_one_minus_alphas[index], g[k + "noise"])
        assert bits_equal(xp.cpu(), oxp) and bits_equal(p0.cpu(), op0)  # kernel == oracle, bit for bit


def test_philox_matches_numpy_reference_and_is_shard_invariant():
    from oracle.philox_ref import philox_randn
    m = _model(10)
    full = m._philox((6, 3, 5, 7), torch.device(DEV), 1234, 0, 7, 1).cpu()
    lo = m._philox((2, 3, 5, 7), torch.device(DEV), 1234, 0, 7, 1).cpu()
    hi = m._philox((4, 3, 5, 7), torch.device(DEV), 1234, 2, 7, 1).cpu()
    assert bits_equal(torch.cat([lo, hi]), full)  # invariant to how samples are split over ranks
    ref = philox_randn(6, 105, 1234, 0, 7, 1).reshape(6, 3, 5, 7)
    assert np.abs(full.numpy() - ref).max() < 2e-5
    big = m._philox((4, 3, 64, 64), torch.device(DEV), 99, 0, 3, 1).cpu()
    assert abs(float(big.mean())) < 0.02 and abs(float(big.std()) - 1.0) < 0.02


def test_ldm_ddpm_tables_and_step_vs_oracle():
    """ddpm.py-flavoured sampler: tables == oracle restatement (same float64 numpy ops), fused p_sample vs oracle."""
    from eo_diffusion_amd.diffusion.ddpm import DDPM

    class Fixed(torch.nn.Module):
        def forward(self, x, t, cond=None, y=None):
            return self.pred

    for sch in ("linear", "cosine"):
        m = DDPM(Fixed(), timesteps=1000, beta_schedule=sch, image_size=8, channels=3).to(DEV)
        lt = SCH.ldm_register_schedule(SCH.ldm_beta_schedule(sch, 1000))
        for k, v in lt.items():
            assert bits_equal(getattr(m, k).cpu(), v), (sch, k)
        x = synth_input("lx", (3, 3, 6, 5), 41)
        e = synth_input("le", (3, 3, 6, 5), 41)
        z = synth_input("lz", (3, 3, 6, 5), 41)
        for tv in ([999, 500, 1], [0, 0, 7], [3, 0, 250]):
            t = torch.tensor(tv)
            m.model.pred = e.to(DEV)
            for clip in (True, False):
                got = m.p_sample(x.to(DEV), t.to(DEV), clip_denoised=clip, noise=z.to(DEV)).cpu()
                ref = SR.ldm_p_sample(lt, x, t, e, z, clip_denoised=clip)
                assert float((got - ref).abs().max()) <= 2e-6 * max(1.0, float(ref.abs().max())), (sch, tv, clip)
        q = m.q_sample(x.to(DEV), torch.tensor([0, 10, 999]).to(DEV), z.to(DEV)).cpu()
        assert bits_equal(q, SR.q_sample(lt, x, torch.tensor([0, 10, 999]), z))


def test_cfg_combine_bit_exact():
    from eo_diffusion_amd import _lib
    eu = synth_input("cu", (2, 3, 9, 7), 42).to(DEV)
    ec = synth_input("cc", (2, 3, 9, 7), 42).to(DEV)
    out = torch.empty_like(eu)
    _lib.check(_lib.lib().eod_cfg_combine(eu.data_ptr(), ec.data_ptr(), 3.5, out.data_ptr(), out.numel(),
                                          torch.cuda.current_stream().cuda_stream))
    ref = eu.cpu() + 3.5 * (ec.cpu() - eu.cpu())
    assert bits_equal(out.cpu(), ref)
