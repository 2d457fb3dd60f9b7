"""Test infrastructure: the electric-fence device allocator (efence_alloc.cpp).  install() must run before the process touches the GPU."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libefence.so")
SRC = os.path.join(HERE, "efence_alloc.cpp")


def build():
    if os.path.exists(SO) and os.path.getmtime(SO) >= os.path.getmtime(SRC):
        return SO
    subprocess.run(["/opt/rocm/bin/hipcc", "-O2", "-shared", "-fPIC", "-std=c++17", "-o", SO, SRC], check=True)
    return SO


def install():
    import torch
    from torch.cuda.memory import CUDAPluggableAllocator, change_current_allocator
    change_current_allocator(CUDAPluggableAllocator(build(), "efence_malloc", "efence_free"))
