"""GPU parity: whole sampling trajectories (DDPM + RePaint, DDIM) with injected noise vs the golden vectors the
reference's own sampling() / DDIMSampler.sample() produced, plus training-forward and sharding invariance."""
import numpy as np
import pytest
import torch

from tests.gpu_util import DEV, load_into
from tests.helpers import bits_equal, gt, rel_l2, unet_cfgs
from tests.synth import synth_input, synth_state_dict

pytestmark = pytest.mark.gpu
TRAJ_TOL = {"fp32": 2e-5, "fp16": 1e-2, "fp32x3": 2e-5}  # 10-20 recursive steps through the UNet (fp32x3: the exact mode's gate)


def _model(prec, T=20, cond_type=None):
    from eo_diffusion_amd.backbones.unet_openai import UNetModel, unet_param_shapes
    from eo_diffusion_amd.diffusion.model import EODiffusion
    cfg = unet_cfgs()["u_a0_tiny"]
    u = UNetModel(**cfg).set_precision(prec)
    u.load_state_dict(synth_state_dict(unet_param_shapes(**cfg), 7))
    return EODiffusion(u, timesteps=T, image_size=16, in_channels=3, cond_type=cond_type, device=DEV).to(DEV).eval()


@pytest.mark.parametrize("prec", ["fp32", "fp16", "fp32x3"])
@pytest.mark.parametrize("name,clip,masked", [("traj_ddpm_repaint_clip_T20", True, True),
                                              ("traj_ddpm_repaint_noclip_T20", False, True),
                                              ("traj_ddpm_uncond_clip_T20", True, False)])
def test_ddpm_trajectory_vs_golden(prec, name, clip, masked):
    g = gt(name)
    m = _model(prec, cond_type="sum" if masked else None)
    cond = torch.cat([g["gt"], g["mask"]], 1).to(DEV) if masked else None
    out = m.sampling(2, clipped_reverse_diffusion=clip, device=DEV, cond=cond, x_T=g["x_T"], noises=g["noises"], progress=False)
    assert rel_l2(out.cpu(), g["out"]) < TRAJ_TOL[prec]


@pytest.mark.parametrize("prec", ["fp32", "fp16", "fp32x3"])
@pytest.mark.parametrize("clip", [True, False])
def test_full_1000_step_sampling_vs_reference_output(prec, clip):
    """EODiffusion.sampling over the reference's real chain length (T = 1000, diffusion/model.py:46-92) against the output of the
    reference's own call (tests/golden/make_golden.py gen_full_chain); the noise is what torch.randn yields after the recorded seed.
    Unclipped, |x_t| reaches 9e4: fp32x3 must follow the reference there (fp32-grade on the whole fp32 domain); fp16 STORAGE cannot
    hold such values (65504), the reference's own fp16 mode cannot either -- that combination is only required to stay loud."""
    from tests.test_oracle_golden import _full_chain_inputs
    g = gt("traj_ddpm_uncond_T1000_full")
    tag = "clip" if clip else "noclip"
    xT, noises = _full_chain_inputs(int(g[tag + "_seed"]))
    m = _model(prec, T=1000)
    out = m.sampling(2, clipped_reverse_diffusion=clip, device=DEV, x_T=xT, noises=noises, progress=False).cpu()
    if prec == "fp16" and not clip:
        assert not torch.isfinite(out).all()   # overflow of the storage type shows up as inf / NaN, never as a plausible image
        return
    err = rel_l2(out, g[tag + "_out"])
    print(f"1000 DDPM steps [{prec}, {tag}], max|out| = {float(g[tag + '_out'].abs().max()):.3g}: rel-L2 vs the reference = {err:.3e}")
    assert err < TRAJ_TOL[prec]


@pytest.mark.parametrize("prec", ["fp32", "fp16", "fp32x3"])
@pytest.mark.parametrize("eta", [0.0, 0.5])
def test_full_ddim250_repaint_call_vs_reference_output(prec, eta):
    """BASELINE config 3's call shape at its real length: DDIMSampler.ddim_sampling with 250 of 1000 steps and the RePaint mask mix
    (inference.py:112-126, ddim.py:56-164) on the attention UNet u_a1_tiny, against the output of the reference's own call"""
    from eo_diffusion_amd.backbones.unet_openai import UNetModel, unet_param_shapes
    from eo_diffusion_amd.diffusion.ddim import DDIMSampler
    from eo_diffusion_amd.diffusion.model import EODiffusion
    from tests.test_oracle_golden import _full_ddim_inputs
    g = gt("traj_ddim_S250_T1000_repaint_full")
    cfg = unet_cfgs()["u_a1_tiny"]
    u = UNetModel(**cfg).set_precision(prec)
    u.load_state_dict(synth_state_dict(unet_param_shapes(**cfg), 7))
    m = EODiffusion(u, timesteps=1000, image_size=16, in_channels=3, device=DEV).to(DEV).eval()
    s = DDIMSampler(m)
    s.make_schedule(ddim_num_steps=250, ddim_eta=eta, verbose=False)
    assert np.array_equal(np.asarray(s.ddim_timesteps, np.int64), g[f"eta{eta}_steps"].numpy())
    xT, stp, mix = _full_ddim_inputs(int(g[f"eta{eta}_seed"]), 250)
    out, inter = s.ddim_sampling(None, (2, 3, 16, 16), x_T=xT, mask=g["mask"], x0=g["x0"], log_every_t=50, step_noises=stp, mix_noises=mix,
                                 progress=False)
    e_out, e_p0 = rel_l2(out.cpu(), g[f"eta{eta}_out"]), rel_l2(inter["pred_x0"][-1].cpu(), g[f"eta{eta}_pred_x0_last"])
    print(f"DDIM 250 of 1000 + RePaint [{prec}, eta {eta}]: rel-L2 vs the reference: out {e_out:.3e}, last pred_x0 {e_p0:.3e}")
    assert e_out < TRAJ_TOL[prec] and e_p0 < TRAJ_TOL[prec]


@pytest.mark.parametrize("prec", ["fp32", "fp16", "fp32x3"])
def test_training_forward_vs_golden(prec):
    g = gt("train_forward_T20")
    m = _model(prec)
    with torch.no_grad():
        x_t = m._forward_diffusion(g["x0"].to(DEV), g["t"].to(DEV), g["noise"].to(DEV))
        pred = m.model(x_t, g["t"].to(DEV))
    assert rel_l2(pred.cpu(), g["pred"]) < (5e-3 if prec == "fp16" else 1e-5)


@pytest.mark.parametrize("prec", ["fp32", "fp16", "fp32x3"])
@pytest.mark.parametrize("tag,S,eta,masked", [("S10_eta0", 10, 0.0, False), ("S10_eta05_mask", 10, 0.5, True),
                                              ("S20_eta0_mask", 20, 0.0, True)])
def test_ddim_trajectory_vs_golden(prec, tag, S, eta, masked):
    from eo_diffusion_amd.diffusion.ddim import DDIMSampler
    g = gt("traj_ddim_" + tag + "_T20")
    m = _model(prec)
    s = DDIMSampler(m)
    s.make_schedule(ddim_num_steps=S, ddim_eta=eta, verbose=False)
    assert np.array_equal(np.asarray(s.ddim_timesteps, np.int64), g["steps"].numpy())  # integer schedule: exact
    out, inter = s.ddim_sampling(None, (2, 3, 16, 16), x_T=g["x_T"], mask=g.get("mask"), x0=g.get("x0"), log_every_t=5,
                                 step_noises=g["step_noises"], mix_noises=g.get("mix_noises"), progress=False)
    assert rel_l2(out.cpu(), g["out"]) < TRAJ_TOL[prec]
    assert rel_l2(inter["pred_x0"][-1].cpu(), g["pred_x0_last"]) < TRAJ_TOL[prec]
    assert len(inter["x_inter"]) == int(g["n_inter"])


def test_philox_sampling_is_invariant_to_batch_sharding():
    """4 samples in one batch == 2 + 2 samples with sample_offset (what two ranks would compute)."""
    m = _model("fp32", T=6)
    full = m.sampling(4, device=DEV, rng="philox", seed=7, progress=False)
    lo = m.sampling(2, device=DEV, rng="philox", seed=7, sample_offset=0, progress=False)
    hi = m.sampling(2, device=DEV, rng="philox", seed=7, sample_offset=2, progress=False)
    assert torch.equal(torch.cat([lo, hi]), full)


def test_ddim_classifier_free_guidance_branch():
    """ddim.py:177-181: doubled batch through the UNet (concat conditioning), then e_u + s*(e_c - e_u)."""
    from eo_diffusion_amd.backbones.unet_openai import UNetModel, unet_param_shapes
    from eo_diffusion_amd.diffusion.ddim import DDIMSampler
    from eo_diffusion_amd.diffusion.model import EODiffusion
    from oracle import sampler_ref as SR
    from oracle import schedule as SCH
    from oracle import unet_ref as UR
    from tests.synth import synth_input
    cfg = dict(image_size=16, in_channels=7, model_channels=32, out_channels=3, num_res_blocks=1, attention_resolutions=[2],
               channel_mult=[1, 2], num_heads=2)
    sd = synth_state_dict(unet_param_shapes(**cfg), 7)
    u = UNetModel(**cfg).set_precision("fp32")
    u.load_state_dict(sd)
    m = EODiffusion(u, timesteps=20, image_size=16, in_channels=3, device=DEV).to(DEV).eval()
    s = DDIMSampler(m)
    s.make_schedule(ddim_num_steps=10, ddim_eta=0.0, verbose=False)
    x = synth_input("gx", (2, 3, 16, 16), 51)
    c = synth_input("gc", (2, 4, 16, 16), 51, uniform=True)
    uc = torch.zeros_like(c)
    nz = synth_input("gn", (2, 3, 16, 16), 51)
    index = 5
    t = torch.full((2,), int(s.ddim_timesteps[index]), dtype=torch.long)
    xp, p0 = s.p_sample_ddim(x.to(DEV), c.to(DEV), t.to(DEV), index=index, unconditional_guidance_scale=2.5,
                             unconditional_conditioning=uc.to(DEV), _noise=nz)
    e_u = UR.unet_forward(sd, cfg, x, t, cond=uc)
    e_c = UR.unet_forward(sd, cfg, x, t, cond=c)
    e = e_u + 2.5 * (e_c - e_u)
    tb = SCH.eo_cosine_tables(20)
    dd = SCH.ddim_tables(tb["alphas_cumprod"], SCH.ddim_timesteps("uniform", 10, 20), 0.0)
    rxp, rp0 = SR.ddim_step(x, e, dd["a"][index], dd["a_prev"][index], dd["sigma"][index], dd["sqrt_1m_a"][index], nz)
    assert rel_l2(xp.cpu(), rxp) < 2e-5 and rel_l2(p0.cpu(), rp0) < 2e-5


@pytest.mark.parametrize("prec", ["fp32", "fp16", "fp32x3"])
def test_full_guided_ddim_call_vs_reference_output(prec):
    """classifier-free guidance through a whole call (ddim.py:177-181): DDIMSampler.ddim_sampling with 50 of 1000 steps, eta 0.3, guidance
    scale 2.5 and concat conditioning, against the output of the reference's own DDIMSampler.sample call"""
    from eo_diffusion_amd.backbones.unet_openai import UNetModel, unet_param_shapes
    from eo_diffusion_amd.diffusion.ddim import DDIMSampler
    from eo_diffusion_amd.diffusion.model import EODiffusion
    from tests.test_oracle_golden import CFG_CFG, _full_cfg_inputs
    g = gt("traj_ddim_S50_T1000_cfg_full")
    S, eta, scale = (float(v) for v in g["hyper"])
    u = UNetModel(**CFG_CFG).set_precision(prec)
    u.load_state_dict(synth_state_dict(unet_param_shapes(**CFG_CFG), 7))
    m = EODiffusion(u, timesteps=1000, image_size=16, in_channels=3, device=DEV).to(DEV).eval()
    s = DDIMSampler(m)
    s.make_schedule(ddim_num_steps=int(S), ddim_eta=eta, verbose=False)
    assert np.array_equal(np.asarray(s.ddim_timesteps, np.int64), g["steps"].numpy())
    xT, stp = _full_cfg_inputs(int(g["seed"]), int(S))
    c = g["cond"].to(DEV)
    out, inter = s.ddim_sampling(c, (2, 3, 16, 16), x_T=xT, log_every_t=10, unconditional_guidance_scale=scale,
                                 unconditional_conditioning=torch.zeros_like(c), step_noises=stp, progress=False)
    e_out, e_p0 = rel_l2(out.cpu(), g["out"]), rel_l2(inter["pred_x0"][-1].cpu(), g["pred_x0_last"])
    print(f"guided DDIM 50 of 1000 [{prec}]: rel-L2 vs the reference: out {e_out:.3e}, last pred_x0 {e_p0:.3e}")
    assert e_out < TRAJ_TOL[prec] and e_p0 < TRAJ_TOL[prec]


def test_ldm_ddpm_loop_with_mask_vs_oracle():
    """DDPM.p_sample_loop (ddpm.py:1296-1345): mask mix AFTER the step with fresh noise."""
    from eo_diffusion_amd.backbones.unet_openai import UNetModel, unet_param_shapes
    from eo_diffusion_amd.diffusion.ddpm import DDPM
    from oracle import sampler_ref as SR
    from oracle import schedule as SCH
    from oracle import unet_ref as UR
    from tests.synth import rect_mask, synth_input
    cfg = unet_cfgs()["u_a0_tiny"]
    sd = synth_state_dict(unet_param_shapes(**cfg), 7)
    u = UNetModel(**cfg).set_precision("fp32")
    u.load_state_dict(sd)
    T = 8
    m = DDPM(u, timesteps=T, beta_schedule="linear", image_size=16, channels=3).to(DEV).eval()
    lt = SCH.ldm_register_schedule(SCH.ldm_beta_schedule("linear", T))
    xT = synth_input("dxT", (2, 3, 16, 16), 61)
    x0 = synth_input("dx0", (2, 3, 16, 16), 61, uniform=True)
    mask = rect_mask(2, 16, 16, 61)
    noises = [synth_input(f"dn{k}", (2, 3, 16, 16), 61) for k in range(T)]
    mixes = [synth_input(f"dm{k}", (2, 3, 16, 16), 61) for k in range(T)]
    out = m.p_sample_loop((2, 3, 16, 16), x_T=xT, mask=mask, x0=x0, noises=noises, mix_noises=mixes).cpu()
    img = xT
    for k, i in enumerate(reversed(range(T))):
        ts = torch.full((2,), i, dtype=torch.long)
        img = SR.ldm_p_sample(lt, img, ts, UR.unet_forward(sd, cfg, img, ts), noises[k])
        img = SR.q_sample(lt, x0, ts, mixes[k]) * mask + (1.0 - mask) * img
    assert rel_l2(out, img) < 2e-5


def test_full_size_sharding_invariance_and_determinism():
    """BASELINE shape family (A0 arch, base 128, 128x128 here to keep the test short): a batch of 4 sampled in one go
    equals 2 + 2 samples with sample_offset (what two ranks compute), bit for bit, and a second run is bit-identical."""
    from eo_diffusion_amd.backbones.unet_openai import UNetModel
    from eo_diffusion_amd.diffusion.model import EODiffusion
    torch.manual_seed(0)
    u = UNetModel(128, in_channels=3, model_channels=128, out_channels=3, num_res_blocks=1, attention_resolutions=[],
                  channel_mult=[1, 2, 3, 4], num_heads=1).set_precision("fp16")
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for p in u.parameters():
            if p.dim() > 1 and float(p.abs().max()) == 0.0:
                p.copy_(torch.randn(p.shape, generator=g) * 0.02)
    m = EODiffusion(u, timesteps=3, image_size=128, in_channels=3, device=DEV).to(DEV).eval()
    full = m.sampling(4, device=DEV, rng="philox", seed=5, progress=False)
    again = m.sampling(4, device=DEV, rng="philox", seed=5, progress=False)
    lo = m.sampling(2, device=DEV, rng="philox", seed=5, sample_offset=0, progress=False)
    hi = m.sampling(2, device=DEV, rng="philox", seed=5, sample_offset=2, progress=False)
    assert torch.isfinite(full).all()
    assert torch.equal(full, again)
    assert torch.equal(torch.cat([lo, hi]), full)


def test_repaint_keeps_known_region_statistics():
    """RePaint cond_type='sum' (model.py:58-60): at the last step (t = 0) the kept region of the UNet input is
    sqrt(acp_0)*gt + sqrt(1-acp_0)*eps, i.e. gt up to the t=0 noise level -- checked through the product loop with a
    denoiser that predicts exactly the injected noise."""
    from eo_diffusion_amd.diffusion.model import EODiffusion
    from tests.synth import rect_mask, synth_input

    class Echo(torch.nn.Module):
        def forward(self, x, t, cond=None, y=None):
            self.last_x = x.clone()
            return torch.zeros_like(x)

    m = EODiffusion(Echo(), timesteps=4, image_size=16, in_channels=3, cond_type="sum", device=DEV).to(DEV)
    gt0 = synth_input("rk", (2, 3, 16, 16), 71, uniform=True).to(DEV)
    mask = rect_mask(2, 16, 16, 71).to(DEV)
    noises = torch.zeros(4, 2, 3, 16, 16)
    m.sampling(2, device=DEV, cond=torch.cat([gt0, mask], 1), x_T=torch.zeros(2, 3, 16, 16), noises=noises, progress=False)
    kept = mask.expand_as(gt0) > 0
    expect = m.sqrt_alphas_cumprod[0] * gt0
    assert torch.allclose(m.model.last_x[kept], expect[kept], rtol=0, atol=1e-6)


def test_hipgraph_replay_is_bit_identical():
    """UNetModel.enable_graph(): the launch program captured into one hipGraph returns the same bits as the plain replay,
    for changing inputs / timesteps (static input buffers are refreshed before every replay)"""
    m = _model("fp16")
    x1, x2 = synth_input("gx1", (2, 3, 16, 16), 1).to(DEV), synth_input("gx2", (2, 3, 16, 16), 2).to(DEV)
    t1, t2 = torch.tensor([3, 17], device=DEV), torch.tensor([19, 0], device=DEV)
    with torch.no_grad():
        ref1, ref2 = m.model(x1, t1).clone(), m.model(x2, t2).clone()
        m.model.enable_graph(True)
        g1 = m.model(x1, t1).clone()
        g2 = m.model(x2, t2).clone()
        g1b = m.model(x1, t1).clone()
        m.model.enable_graph(False)
    assert bits_equal(g1, ref1) and bits_equal(g2, ref2) and bits_equal(g1b, ref1)


def test_hipgraph_capture_after_a_fractional_timestep_call():
    """a call with floating-point timesteps runs un-captured and binds the timestep kernel to the fp32 slot; the first capture afterwards
    (integer timesteps) must bind it back, or the graph would read the int64 words as fp32 for its whole life"""
    m = _model("fp16")
    x1 = synth_input("gx1", (2, 3, 16, 16), 1).to(DEV)
    t1 = torch.tensor([3, 17], device=DEV)
    with torch.no_grad():
        ref1 = m.model(x1, t1).clone()
        m.model.enable_graph(True)
        reff = m.model(x1, torch.tensor([2.5, 16.25], device=DEV)).clone()   # un-captured
        g1 = m.model(x1, t1).clone()                                          # first capture
        g1b = m.model(x1, t1).clone()                                         # replay
        m.model.enable_graph(False)
    assert bits_equal(g1, ref1) and bits_equal(g1b, ref1) and not bits_equal(reff, ref1)
