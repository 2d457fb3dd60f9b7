"""Helpers for the -m gpu parity tests (HIP path vs CPU oracle / golden fixtures)."""
import torch

from eo_diffusion_amd import _lib
from eo_diffusion_amd.engine import Program, round_up

DEV = "cuda:0"
# stated tolerances (rel-L2 vs the fp32 CPU oracle): SURVEY.md section 8c noise-floor measurements
# ("fp32x3": fp32 storage, 3x3 convs through three fp16 MFMAs on split operands -- held to the SAME gate as exact fp32)
TOL = {"fp32": 1e-5, "fp16": 5e-3, "fp32x3": 1e-5}


def run_program(prec, x_nchw, emit):
    """x NCHW fp32 (cpu) -> NHWC storage -> emit(prog, act) -> NCHW fp32 (cpu)."""
    x = x_nchw.to(DEV).float().contiguous()
    N, C, H, W = x.shape
    prog = Program(DEV, prec)
    cp = round_up(C, prog.epc)
    a, idx = prog.to_nhwc(N, C, 0, H, W, cp)
    prog.ops[idx].u.small.p[0] = x.data_ptr()
    y = emit(prog, a)
    out = torch.empty((y.N, y.C, y.H, y.W), dtype=torch.float32, device=DEV)
    i2 = prog.to_nchw(y)
    prog.ops[i2].u.small.p[1] = out.data_ptr()
    prog.run()
    torch.cuda.synchronize()
    return out.cpu()


def load_into(module, sd):
    missing = module.load_state_dict(sd, strict=True)
    return module.to(DEV).eval()


def program_empty_at_segment_end(self, shape, dtype=None, zero=False):
    """test-side replacement of engine.Program.empty (monkeypatched in by the `tail_alloc` fixture): every program buffer ENDS where its
    own allocator segment ends (requests of 10 MiB and more get a segment of exactly their rounded size from torch's caching allocator),
    so that a kernel reading or writing past the logical end of a buffer leaves the mapped range and faults on EVERY run instead of on
    the rare layout where the buffer happens to be the last one of a segment (the per-sample bias read of round 3)."""
    import math
    dtype = dtype or self.tdtype
    shape = (int(shape),) if isinstance(shape, int) else tuple(int(v) for v in shape)
    esz = torch.empty((), dtype=dtype).element_size()
    nbytes = max(16, int(math.prod(shape)) * esz)
    seg = ((nbytes + (2 << 20) - 1) // (2 << 20)) * (2 << 20) + (10 << 20)
    base = torch.empty((seg,), dtype=torch.uint8, device=self.device)
    self.keep.append(base)
    start = (seg - nbytes) & ~15
    t = base[start:start + int(math.prod(shape)) * esz].view(dtype).view(shape)
    if zero:
        t.zero_()
    self.keep.append(t)
    self.nbytes += seg
    return t
