"""CPU oracle for the EODiffusion hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

This package is a plain-PyTorch-CPU / numpy restatement of the reference algorithm
(furio1999/EO_Diffusion: backbones/unet_openai.py, diffusion/model.py, diffusion/ddim.py,
diffusion/util.py and the sampler algebra of diffusion/ddpm.py).  It exists only so that
`tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` can check /
time the HIP path against it.  Nothing under `eo_diffusion_amd/` may import it.

Parity pin: the reference ships no tests and no golden vectors, so the
oracle is pinned against outputs of the reference itself, generated in the build container
by importing /root/reference (script: tests/golden/make_golden.py, fixtures: tests/golden/*.npz).
`tests/test_oracle_golden.py` replays every fixture through this package.
"""
