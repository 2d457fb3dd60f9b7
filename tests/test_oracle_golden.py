"""CPU suite: pin the oracle (oracle/) against the golden vectors generated from the reference
(tests/golden/make_golden.py).  Integer schedules exact; fp32 sampler algebra bit-exact; the UNet
within 1e-6 rel-L2 (same torch CPU ops, so in practice exact)."""
import numpy as np
import pytest
import torch

from oracle import sampler_ref as SR
from oracle import schedule as SCH
from oracle import unet_ref as UR
from tests.helpers import bits_equal, close_ulp, gload, gt, key_contracts, rel_l2, unet_cfgs
from tests.synth import synth_input, synth_state_dict

torch.set_num_threads(8)


@pytest.mark.parametrize("T", [20, 200, 1000])
def test_cosine_tables_bit_exact(T):
    g = gt(f"schedule_T{T}")
    tb = SCH.eo_cosine_tables(T)
    for k, v in g.items():
        if k.startswith("sqrt_"):  # torch CPU sqrt is CPU-model dependent (<= 1 ulp), see sampler_ref._sqrt
            assert close_ulp(tb[k], v), k
        else:
            assert bits_equal(tb[k], v), k


@pytest.mark.parametrize("sch", ["linear", "cosine", "sqrt_linear", "sqrt"])
def test_ldm_betas(sch):
    g = gload(f"ldm_betas_{sch}_T1000")["betas"]
    b = SCH.ldm_beta_schedule(sch, 1000)
    assert b.dtype == np.float64 and np.array_equal(b, g)


@pytest.mark.parametrize("T,S", [(1000, 50), (1000, 250), (1000, 600), (1000, 1000), (20, 10), (20, 20)])
@pytest.mark.parametrize("eta", [0.0, 0.5])
def test_ddim_tables(T, S, eta):
    g = gload(f"ddim_S{S}_T{T}_eta{eta}")
    steps = SCH.ddim_timesteps("uniform", S, T)
    assert steps.dtype == np.int64 and np.array_equal(steps, g["steps"])  # integer schedule: exact
    dd = SCH.ddim_tables(SCH.eo_cosine_tables(T)["alphas_cumprod"], steps, eta)
    assert np.array_equal(np.asarray(dd["a"]), g["a"])
    assert np.array_equal(np.asarray(dd["a_prev"], dtype=np.float64), g["a_prev"])
    assert np.allclose(np.asarray(torch.as_tensor(dd["sigma"]).double()), g["sigma"], rtol=1e-12, atol=0)
    assert close_ulp(torch.as_tensor(np.asarray(dd["sqrt_1m_a"])), torch.as_tensor(g["sqrt_1m_a"]))


def test_ddim_known_values():
    # S=250,T=1000 -> [1,5,...,997]; S in (500,1000] -> c=1 -> 1000 steps, shifted by -1
    s = SCH.ddim_timesteps("uniform", 250, 1000)
    assert s[0] == 1 and s[1] == 5 and s[-1] == 997 and len(s) == 250
    s = SCH.ddim_timesteps("uniform", 600, 1000)
    assert len(s) == 1000 and s[0] == 0 and s[-1] == 999
    assert np.array_equal(SCH.ddim_timesteps("quad", 50, 1000), gload("ddim_quad_S50_T1000")["steps"])


def test_timestep_embedding():
    g = gt("temb")
    for d in (32, 128, 33):
        assert bits_equal(UR.timestep_embedding(g["t"], d), g[f"d{d}"])


def _sd_for(prefix_shapes, seed):
    return synth_state_dict(prefix_shapes, seed)


RES = {
    "res_same": (32, 32, {}), "res_change": (32, 64, {}), "res_cat96": (96, 32, {}),
    "res_skip3x3": (32, 64, {"skip3": True}), "res_film": (64, 32, {"film": True}),
    "res_up": (32, 32, {"up": True}), "res_down": (32, 64, {"down": True}),
}


def res_shapes(cin, cout, film=False, skip3=False, emb=128):
    s = {"in_layers.0.weight": (cin,), "in_layers.0.bias": (cin,), "in_layers.2.weight": (cout, cin, 3, 3),
         "in_layers.2.bias": (cout,), "emb_layers.1.weight": ((2 if film else 1) * cout, emb),
         "emb_layers.1.bias": ((2 if film else 1) * cout,), "out_layers.0.weight": (cout,),
         "out_layers.0.bias": (cout,), "out_layers.3.weight": (cout, cout, 3, 3), "out_layers.3.bias": (cout,)}
    if cin != cout:
        k = 3 if skip3 else 1
        s["skip_connection.weight"] = (cout, cin, k, k)
        s["skip_connection.bias"] = (cout,)
    return s


@pytest.mark.parametrize("name", list(RES))
def test_resblock(name):
    cin, cout, kw = RES[name]
    g = gt("mod_" + name)
    sd = synth_state_dict(res_shapes(cin, cout, kw.get("film", False), kw.get("skip3", False)), 3)
    sd = {"b." + k: v for k, v in sd.items()}
    y = UR.res_block(sd, "b", g["x"], g["emb"], film=kw.get("film", False), up=kw.get("up", False), down=kw.get("down", False))
    assert rel_l2(y, g["y"]) < 1e-6


ATT = {
    "attn_c64_h1_legacy": (64, 1, False), "attn_c128_h4_legacy": (128, 4, False), "attn_c384_h8_legacy": (384, 8, False),
    "attn_c128_d128_new": (128, 1, True), "attn_c128_h8_new": (128, 8, True), "attn_c512_h8_legacy": (512, 8, False),
}


def attn_shapes(C):
    return {"norm.weight": (C,), "norm.bias": (C,), "qkv.weight": (3 * C, C, 1), "qkv.bias": (3 * C,),
            "proj_out.weight": (C, C, 1), "proj_out.bias": (C,)}


@pytest.mark.parametrize("name", list(ATT))
def test_attention(name):
    C, heads, new = ATT[name]
    g = gt("mod_" + name)
    sd = {"a." + k: v for k, v in synth_state_dict(attn_shapes(C), 4).items()}
    y = UR.attention_block(sd, "a", g["x"], heads, new)
    assert rel_l2(y, g["y"]) < 1e-6


def test_resample():
    conv = lambda p: {p + ".weight": (32, 32, 3, 3), p + ".bias": (32,)}
    for name in ("up_conv", "up_conv_3x3"):
        g = gt("mod_" + name)
        sd = {"u." + k: v for k, v in synth_state_dict(conv("conv"), 5).items()}
        assert rel_l2(UR.upsample(sd, "u", g["x"]), g["y"]) < 1e-6
    g = gt("mod_up_noconv")
    assert bits_equal(UR.upsample({}, "", g["x"], use_conv=False), g["y"])
    for name in ("down_conv", "down_conv_odd"):
        g = gt("mod_" + name)
        sd = {"d." + k: v for k, v in synth_state_dict(conv("op"), 5).items()}
        assert rel_l2(UR.downsample(sd, "d", g["x"]), g["y"]) < 1e-6
    g = gt("mod_down_pool")
    assert rel_l2(UR.downsample({}, "", g["x"], use_conv=False), g["y"]) < 1e-7


@pytest.mark.parametrize("name", list(unet_cfgs()))
def test_unet_forward(name):
    cfg = unet_cfgs()[name]
    g = gt("unet_" + name)
    # the key/shape contract for this cfg comes from the product's own shape walker, which is
    # itself checked against the reference's key list in test_host_contract.py
    from eo_diffusion_amd.backbones.unet_openai import unet_param_shapes
    sd = synth_state_dict(unet_param_shapes(**cfg), 7)
    y = UR.unet_forward(sd, cfg, g["x"], g["t"], cond=g.get("cond"), y=g.get("y"))
    assert y.shape == g["y_out"].shape
    assert rel_l2(y, g["y_out"]) < 1e-6


@pytest.mark.parametrize("tag", ["t999", "t500", "t1", "t0", "tmix0", "tmix"])
def test_sampler_steps_vs_golden(tag):
    g = gt("sampler_steps_T1000")
    tb = SCH.eo_cosine_tables(1000)
    t, x, pred, noise = g[tag + "_t"], g[tag + "_x"], g[tag + "_pred"], g[tag + "_noise"]
    # golden = the reference's torch CPU result, whose sqrt is off by <= 1 ulp in a CPU-dependent way
    assert close_ulp(SR.ddpm_step_clip(tb, x, t, noise, pred), g[tag + "_clip"])
    assert close_ulp(SR.ddpm_step_noclip(tb, x, t, noise, pred), g[tag + "_noclip"])
    assert close_ulp(SR.q_sample(tb, x, t, noise), g[tag + "_q"])


def _tiny_eps():
    cfg = unet_cfgs()["u_a0_tiny"]
    from eo_diffusion_amd.backbones.unet_openai import unet_param_shapes
    sd = synth_state_dict(unet_param_shapes(**cfg), 7)
    return lambda x, t: UR.unet_forward(sd, cfg, x, t)


@pytest.mark.parametrize("name,clip,masked", [("traj_ddpm_repaint_clip_T20", True, True),
                                              ("traj_ddpm_repaint_noclip_T20", False, True),
                                              ("traj_ddpm_uncond_clip_T20", True, False)])
def test_ddpm_trajectory(name, clip, masked):
    g = gt(name)
    tb = SCH.eo_cosine_tables(20)
    out = SR.ddpm_sampling(tb, _tiny_eps(), g["x_T"], g["noises"], 20, clip=clip,
                           gt=g["gt"] if masked else None, mask=g["mask"] if masked else None)
    assert rel_l2(out, g["out"]) < 1e-6


def _full_chain_inputs(seed, T=1000, shape=(2, 3, 16, 16)):
    """x_T and the per-step noise of the reference's sampling() call behind traj_ddpm_uncond_T1000_full (one randn per draw, in its
    order; tests/golden/make_golden.py gen_full_chain checked this equality against the recorded draws)"""
    torch.manual_seed(seed)
    xT = torch.randn(shape)
    return xT, [torch.randn(shape) for _ in range(T)]


def test_full_1000_step_clipped_chain_vs_reference_output():
    """the oracle through the reference's COMPLETE 1000-step sampling() call (diffusion/model.py:46-92), output vs the reference's"""
    g = gt("traj_ddpm_uncond_T1000_full")
    xT, noises = _full_chain_inputs(int(g["clip_seed"]))
    with torch.no_grad():
        out = SR.ddpm_sampling(SCH.eo_cosine_tables(1000), _tiny_eps(), xT, noises, 1000, clip=True)
    assert rel_l2(out, g["clip_out"]) < 2e-6


def test_train_forward():
    g = gt("train_forward_T20")
    tb = SCH.eo_cosine_tables(20)
    x_t = SR.q_sample(tb, g["x0"], g["t"], g["noise"])
    assert rel_l2(_tiny_eps()(x_t, g["t"]), g["pred"]) < 1e-6


@pytest.mark.parametrize("tag,S,eta,masked", [("S10_eta0", 10, 0.0, False), ("S10_eta05_mask", 10, 0.5, True),
                                              ("S20_eta0_mask", 20, 0.0, True)])
def test_ddim_trajectory(tag, S, eta, masked):
    g = gt("traj_ddim_" + tag + "_T20")
    tb = SCH.eo_cosine_tables(20)
    steps = SCH.ddim_timesteps("uniform", S, 20)
    assert np.array_equal(steps, g["steps"].numpy())
    dd = SCH.ddim_tables(tb["alphas_cumprod"], steps, eta)
    out, p0 = SR.ddim_sampling(tb, dd, steps, _tiny_eps(), g["x_T"], g["step_noises"],
                               x0=g.get("x0"), mask=g.get("mask"), mix_noises=g.get("mix_noises"))
    assert rel_l2(out, g["out"]) < 1e-6
    assert rel_l2(p0, g["pred_x0_last"]) < 1e-6


def _full_ddim_inputs(seed, nsteps, shape=(2, 3, 16, 16)):
    """x_T, per-step mix noise and per-step DDIM noise of the reference's DDIMSampler.sample call behind
    traj_ddim_S250_T1000_repaint_full: after the seed, x_T, then per step [mix noise, an unused randn_like (ddim.py:171), step noise]"""
    torch.manual_seed(seed)
    xT = torch.randn(shape)
    mix, stp = [], []
    for _ in range(nsteps):
        mix.append(torch.randn(shape))
        torch.randn(shape)
        stp.append(torch.randn(shape))
    return xT, torch.stack(stp), torch.stack(mix)


def _a1_tiny_eps():
    cfg = unet_cfgs()["u_a1_tiny"]
    from eo_diffusion_amd.backbones.unet_openai import unet_param_shapes
    sd = synth_state_dict(unet_param_shapes(**cfg), 7)
    return lambda x, t: UR.unet_forward(sd, cfg, x, t)


@pytest.mark.parametrize("eta", [0.0, 0.5])
def test_full_ddim250_repaint_call_vs_reference_output(eta):
    """the oracle through BASELINE config 3's call shape -- DDIMSampler.sample with 250 of 1000 steps and the RePaint mask mix
    (inference.py:112-126, ddim.py:56-164) on an attention UNet -- against the output of the reference's own call"""
    g = gt("traj_ddim_S250_T1000_repaint_full")
    tb = SCH.eo_cosine_tables(1000)
    steps = SCH.ddim_timesteps("uniform", 250, 1000)
    assert np.array_equal(steps, g[f"eta{eta}_steps"].numpy())
    dd = SCH.ddim_tables(tb["alphas_cumprod"], steps, eta)
    xT, stp, mix = _full_ddim_inputs(int(g[f"eta{eta}_seed"]), len(steps))
    with torch.no_grad():
        out, p0 = SR.ddim_sampling(tb, dd, steps, _a1_tiny_eps(), xT, stp, x0=g["x0"], mask=g["mask"], mix_noises=mix)
    assert rel_l2(out, g[f"eta{eta}_out"]) < 5e-6
    assert rel_l2(p0, g[f"eta{eta}_pred_x0_last"]) < 5e-6


CFG_CFG = dict(image_size=16, in_channels=7, model_channels=32, out_channels=3, num_res_blocks=1, attention_resolutions=[2], channel_mult=[1, 2],
               num_heads=2)


def _full_cfg_inputs(seed, nsteps, shape=(2, 3, 16, 16)):
    """x_T and the step noise of the reference's guided DDIMSampler.sample call: after the seed, x_T, then per step [unused, step noise]"""
    torch.manual_seed(seed)
    xT = torch.randn(shape)
    stp = []
    for _ in range(nsteps):
        torch.randn(shape)
        stp.append(torch.randn(shape))
    return xT, torch.stack(stp)


def test_full_guided_ddim_call_vs_reference_output():
    """classifier-free guidance through a whole DDIMSampler.sample call (ddim.py:177-181: the doubled batch, e_u + s (e_c - e_u)), 50 of
    1000 steps, eta 0.3, scale 2.5, concat conditioning: the oracle against the output of the reference's own call"""
    from eo_diffusion_amd.backbones.unet_openai import unet_param_shapes
    g = gt("traj_ddim_S50_T1000_cfg_full")
    S, eta, scale = (float(v) for v in g["hyper"])
    sd = synth_state_dict(unet_param_shapes(**CFG_CFG), 7)
    c = g["cond"]
    uc = torch.zeros_like(c)

    def eps(x, t):
        e_u, e_c = UR.unet_forward(sd, CFG_CFG, x, t, cond=uc), UR.unet_forward(sd, CFG_CFG, x, t, cond=c)
        return e_u + scale * (e_c - e_u)

    tb = SCH.eo_cosine_tables(1000)
    steps = SCH.ddim_timesteps("uniform", int(S), 1000)
    assert np.array_equal(steps, g["steps"].numpy())
    dd = SCH.ddim_tables(tb["alphas_cumprod"], steps, eta)
    xT, stp = _full_cfg_inputs(int(g["seed"]), len(steps))
    with torch.no_grad():
        out, p0 = SR.ddim_sampling(tb, dd, steps, eps, xT, stp)
    assert rel_l2(out, g["out"]) < 5e-6 and rel_l2(p0, g["pred_x0_last"]) < 5e-6


@pytest.mark.parametrize("eta", [0.0, 0.7])
def test_ddim_single_steps_vs_golden(eta):
    g = gt("ddim_steps_S250_T1000")
    tb = SCH.eo_cosine_tables(1000)
    steps = SCH.ddim_timesteps("uniform", 250, 1000)
    dd = SCH.ddim_tables(tb["alphas_cumprod"], steps, eta)
    for index in (249, 100, 1, 0):
        k = f"eta{eta}_i{index}_"
        xp, p0 = SR.ddim_step(g[k + "x"], g[k + "e"], dd["a"][index], dd["a_prev"][index], dd["sigma"][index],
                              dd["sqrt_1m_a"][index], g[k + "noise"])
        assert close_ulp(xp, g[k + "x_prev"]) and close_ulp(p0, g[k + "pred_x0"])


@pytest.mark.parametrize("sch,T", [("linear", 1000), ("cosine", 1000), ("linear", 20), ("sqrt_linear", 50)])
def test_ldm_register_schedule_vs_harness_derived_tables(sch, T):
    """a21: the 11 register_schedule tables (ddpm.py:122-162).  The fixture is HARNESS-DERIVED (ddpm.py is not importable): betas
    from the reference's importable util.make_beta_schedule, the table formulas of ddpm.py:129-162 evaluated by
    tests/golden/make_golden.py in float64.  The oracle restatement reproduces it bit for bit."""
    g = gt(f"ldm_tables_{sch}_T{T}")
    lt = SCH.ldm_register_schedule(SCH.ldm_beta_schedule(sch, T))
    assert len(g) == 12 and set(g) == set(lt)
    for k, v in g.items():
        assert bits_equal(lt[k], v), (sch, T, k)


def test_ldm_p_sample_reproduces_reference_steps_given_model_py_tables():
    """a23/a24: ddpm.py's p_sample (predict_start_from_noise, clamp, q_posterior, masked noise) is model.py's clipped step
    written on coefficient TABLES.  Fed the tables that model.py's own fp32 expressions give (tests/helpers.eo_tables_as_ldm),
    the restatement reproduces the outputs the REFERENCE produced (sampler_steps_T1000, generated from model.py) to fp32
    rounding -- this pins the step algebra of the un-importable file to reference outputs."""
    from tests.helpers import eo_tables_as_ldm
    g = gt("sampler_steps_T1000")
    lt = eo_tables_as_ldm(gt("schedule_T1000"))
    for tag in ("t999", "t500", "t1", "t0"):
        t, x, pred, noise = g[tag + "_t"], g[tag + "_x"], g[tag + "_pred"], g[tag + "_noise"]
        out = SR.ldm_p_sample(lt, x, t, pred, noise, clip_denoised=True)
        assert rel_l2(out, g[tag + "_clip"]) < 1e-6, tag


def test_ldm_p_sample_matches_eo_step_given_same_tables():
    # ddpm.py's p_sample is algebraically the clipped EO step; pinned only through
    # util.make_beta_schedule (ddpm.py itself is not importable: un-vendored ldm.* / lightning).
    lt = SCH.ldm_register_schedule(SCH.ldm_beta_schedule("linear", 1000))
    tb = {"betas": lt["betas"], "alphas": 1.0 - lt["betas"], "alphas_cumprod": lt["alphas_cumprod"]}
    g = gt("sampler_steps_T1000")
    for tag in ("t999", "t500", "t1", "t0"):
        t, x, pred, noise = g[tag + "_t"], g[tag + "_x"], g[tag + "_pred"], g[tag + "_noise"]
        a = SR.ldm_p_sample(lt, x, t, pred, noise)
        b = SR.ddpm_step_clip(tb, x, t, noise, pred)
        # fp32 `1 - acp` cancels at small t (beta_0/(1-acp_0) = 0.9998 in fp32 vs 1.0 from the float64 tables)
        assert rel_l2(a, b) < (5e-4 if tag in ('t0', 't1') else 2e-5)


def test_philox_known_answer():
    """Random123 known-answer vectors for Philox4x32-10 (Salmon et al.)."""
    from oracle.philox_ref import _philox
    z = np.zeros(1, np.uint32)
    assert [int(x[0]) for x in _philox(z, z, z, z, 0, 0)] == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    f = np.full(1, 0xFFFFFFFF, np.uint32)
    assert [int(x[0]) for x in _philox(f, f, f, f, 0xFFFFFFFF, 0xFFFFFFFF)] == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]


# ---------------------------------------------------------------------------------------------------------------
# optimizer side of the training step (oracle/train_ref.py): pinned against the torch build of the container
# ---------------------------------------------------------------------------------------------------------------
def test_train_ref_adamw_vs_torch():
    import numpy as np
    from oracle import train_ref as TR
    from tests.synth import synth_input
    p0 = synth_input("adam_p", (257,), 1)
    p = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([p], lr=3e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2)
    pn, m, v = p0.numpy().copy(), np.zeros(257, np.float32), np.zeros(257, np.float32)
    for step in range(1, 7):
        g = synth_input(f"adam_g{step}", (257,), 2, scale=0.1)
        p.grad = g.clone()
        opt.step()
        pn, m, v = TR.adamw_step(pn, g.numpy(), m, v, lr=3e-3, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=1e-2, step=step)
        assert np.allclose(pn, p.detach().numpy(), rtol=2e-6, atol=1e-8), step


def test_train_ref_mse_and_ema_vs_torch():
    import numpy as np
    from oracle import train_ref as TR
    from tests.synth import synth_input
    a, b = synth_input("mse_a", (2, 3, 8, 8), 1), synth_input("mse_b", (2, 3, 8, 8), 2)
    ar = a.clone().requires_grad_(True)
    loss = torch.nn.functional.mse_loss(ar, b)
    loss.backward()
    l, d = TR.mse_loss(a.numpy(), b.numpy())
    assert abs(float(l) - float(loss)) < 1e-6 * float(loss)
    assert np.allclose(d, ar.grad.numpy(), rtol=1e-6, atol=1e-10)
    avg = TR.ema_update(a.numpy(), b.numpy(), 0.99)
    assert np.allclose(avg, (0.99 * a + (1 - 0.99) * b).numpy(), rtol=1e-6, atol=1e-8)


@pytest.mark.parametrize("name", ["u_a0_tiny", "u_a1_tiny", "u_film_updown", "u_cond_cls", "u_mnist", "u_s2_13ch"])
def test_oracle_training_gradients_vs_reference(name):
    """torch autograd through the oracle UNet + MSE loss (train.py:116-118) vs the gradients of the REFERENCE itself
    (tests/golden/train_grads_*.npz: per-parameter L2 norm and projection on a fixed direction): pins the training oracle"""
    import json
    import os
    from oracle import unet_ref as UR
    from tests.helpers import GOLDEN, gload, unet_cfgs
    from tests.synth import synth_input, synth_state_dict
    from eo_diffusion_amd.backbones.unet_openai import unet_param_shapes
    g = gload("train_grads_" + name)
    keys = json.load(open(os.path.join(GOLDEN, f"train_grads_{name}_keys.json")))
    cfg = unet_cfgs()[name]
    sd = {k: v.clone().requires_grad_(True) for k, v in synth_state_dict(unet_param_shapes(**cfg), 7).items()}
    tt = lambda k: torch.from_numpy(g[k])
    pred = UR.unet_forward(sd, cfg, tt("x"), tt("t"), cond=tt("cond") if "cond" in g else None, y=tt("y") if "y" in g else None)
    loss = torch.nn.functional.mse_loss(pred, tt("noise"))
    loss.backward()
    assert abs(float(loss.detach()) - float(g["loss"][0])) < 1e-5 * float(g["loss"][0])
    gmax = float(g["grad_norm"].max())
    for i, k in enumerate(keys):
        ref_n, ref_d = float(g["grad_norm"][i]), float(g["grad_dot"][i])
        gr = sd[k].grad.double().flatten()
        direction = synth_input("dir:" + k, (gr.numel(),), 5).double()
        if ref_n < 1e-5 * gmax:
            assert float(gr.norm()) < 1e-3 * gmax, k
            continue
        assert abs(float(gr.norm()) - ref_n) < 1e-4 * ref_n, k
        assert abs(float((gr * direction).sum()) - ref_d) < 1e-4 * ref_n * float(direction.norm()), k


# ------------------------------------------------------------------------------------------------ KeyframeLR (train.py:76-85)
KEYFRAME_CASES = {
    "train_py": dict(units="steps", end=120, lr=1e-3, posmax=30),
    "percent_shorthand": dict(units="percent", end=50, frames=[(0.1, 0.01), "cos", {"position": 0.6, "lr": 0.002}, {"position": "end", "lr": 1e-4}]),
    "implicit_ramps": dict(units="steps", end=40, frames=[{"position": 5, "lr": 0.1}, {"position": 20, "lr": 0.05}]),
    "edge_transitions": dict(units="percent", end=30, frames=["cos", (0.5, 1.0), "linear"]),
}


def _keyframe_frames(name, case):
    import copy
    import math
    if name == "train_py":  # the frame list train.py builds (cos warm-up from lr/100, then lr * exp(-3 * progress))
        lr, posmax, end = case["lr"], case["posmax"], case["end"]
        return [{"position": 0, "lr": lr / 100}, {"transition": "cos"}, {"position": posmax, "lr": lr},
                {"transition": lambda last_lr, sf, ef, pos, *_: lr * math.exp(-3 * (pos - posmax) / (end - posmax))}]
    return copy.deepcopy(case["frames"])


@pytest.mark.parametrize("name", list(KEYFRAME_CASES))
def test_keyframe_lr_reproduces_the_reference_schedule(name):
    """eo_diffusion_amd.train_utils.KeyframeLR against learning rates the REFERENCE's scheduler produced (script_utils/train_utils.py
    :17-226 run by tests/golden/make_golden.py keyframe_lr): every step of the schedule plus two behind its end (the reference keeps the
    last rate there), and sample_lrs(25); train.py's own frame list and the shorthand / implicit-ramp / edge-transition forms"""
    from eo_diffusion_amd.train_utils import KeyframeLR
    g = gload("keyframe_lr")
    case = KEYFRAME_CASES[name]
    opt = torch.optim.AdamW([torch.nn.Parameter(torch.zeros(1))], lr=1e-3)
    sch = KeyframeLR(optimizer=opt, units=case["units"], frames=_keyframe_frames(name, case), end=case["end"])
    lrs = []
    for _ in range(case["end"] + 2):
        lrs.append(opt.param_groups[0]["lr"])
        opt.step()
        sch.step()
    ref = np.asarray(g[name + "_lrs"])
    assert len(lrs) == len(ref)
    assert np.allclose(np.asarray(lrs), ref, rtol=1e-14, atol=0.0), float(np.abs(np.asarray(lrs) - ref).max())
    sch2 = KeyframeLR(optimizer=torch.optim.AdamW([torch.nn.Parameter(torch.zeros(1))], lr=1e-3), units=case["units"],
                      frames=_keyframe_frames(name, case), end=case["end"])
    assert np.allclose(np.asarray(sch2.sample_lrs(25)), np.asarray(g[name + "_sample25"]), rtol=1e-14, atol=0.0)
    assert sch2.last_lr == 0


def test_keyframe_lr_rejects_what_the_reference_rejects():
    from eo_diffusion_amd.train_utils import KeyframeLR
    mk = lambda: torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=0.1)
    with pytest.raises(AssertionError):   # positions must not decrease (train_utils.py:100-102)
        KeyframeLR(mk(), frames=[(0.5, 1.0), (0.2, 0.5)], end=10)
    with pytest.raises(AssertionError):   # nor lie behind the end (:103-105)
        KeyframeLR(mk(), frames=[(0, 1.0), (12, 0.5)], end=10, units="steps")
    with pytest.raises(ValueError):       # unknown transition name (:144)
        s = KeyframeLR(mk(), frames=[(0, 1.0), "cubic", (1.0, 0.5)], end=10)
        s.get_lr_at_pos(0.5)


def test_training_loop_vs_reference_run():
    """12 steps of the reference's training loop (train.py:70-124: EODiffusion.forward, MSELoss, AdamW, KeyframeLR, EMA), re-created from
    oracle pieces -- autograd through oracle/unet_ref.py, oracle/train_ref.py's AdamW / EMA arithmetic, the product's host-side
    KeyframeLR -- against the losses, learning rates and final predictions of the reference's own run (tests/golden/make_golden.py
    gen_train_loop)"""
    import math
    import torch.nn.functional as F
    from eo_diffusion_amd.backbones.unet_openai import unet_param_shapes
    from eo_diffusion_amd.train_utils import KeyframeLR
    from oracle import train_ref as TR
    g = gt("train_loop_12steps_u_a1_tiny")
    steps, lr, posmax, decay = (float(v) for v in g["hyper"])
    steps, posmax = int(steps), int(posmax)
    cfg = unet_cfgs()["u_a1_tiny"]
    sd = {k: v.clone() for k, v in synth_state_dict(unet_param_shapes(**cfg), 7).items()}
    dummy = torch.optim.AdamW([torch.nn.Parameter(torch.zeros(1))], lr=lr)  # carries the learning rate KeyframeLR drives
    sched = KeyframeLR(optimizer=dummy, units="steps", frames=[
        {"position": 0, "lr": lr / 100}, {"transition": "cos"}, {"position": posmax, "lr": lr},
        {"transition": lambda last_lr, sf, ef, pos, *_: lr * math.exp(-3 * (pos - posmax) / (steps - posmax))}], end=steps)
    tb = SCH.eo_cosine_tables(1000)
    mom = {}
    ema = None
    torch.manual_seed(int(g["seed"]))
    losses = []
    for j in range(steps):
        image = synth_input(f"tl_img{j}", (4, 3, 16, 16), 20 + j, uniform=True)
        cur_lr = dummy.param_groups[0]["lr"]
        assert abs(cur_lr - float(g["lrs"][j])) <= 1e-12 * max(1.0, cur_lr)
        noise = torch.randn_like(image)
        t = torch.randint(0, 1000, (4,))
        assert torch.equal(t, g["t"][j])
        sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        pred = UR.unet_forward(sdg, cfg, SR.q_sample(tb, image, t, noise), t)
        loss = F.mse_loss(pred, noise)
        loss.backward()
        losses.append(float(loss.detach()))
        for k, p in sdg.items():
            if p.grad is None:   # the dead nout / conv_out head: AdamW skips parameters without a gradient
                continue
            m_, v_ = mom.setdefault(k, (np.zeros(p.shape, np.float32), np.zeros(p.shape, np.float32)))
            newp, m2, v2 = TR.adamw_step(sd[k].numpy(), p.grad.numpy(), m_, v_, lr=cur_lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.01,
                                         step=j + 1)
            sd[k], mom[k] = torch.from_numpy(np.asarray(newp)), (m2, v2)
        dummy.step()
        sched.step()
        ema = {k: v.clone() for k, v in sd.items()} if ema is None else {k: torch.from_numpy(np.asarray(TR.ema_update(ema[k].numpy(), sd[k].numpy(), decay)))
                                                                       for k in sd}
    ref = g["losses"].numpy()
    assert np.max(np.abs(np.asarray(losses) - ref) / ref) < 1e-4, (losses, ref.tolist())
    with torch.no_grad():
        pm, pe = UR.unet_forward(sd, cfg, g["probe_x"], g["probe_t"]), UR.unet_forward(ema, cfg, g["probe_x"], g["probe_t"])
    assert rel_l2(pm, g["probe_pred_model"]) < 1e-3 and rel_l2(pe, g["probe_pred_ema"]) < 1e-3
