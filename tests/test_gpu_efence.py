"""GPU: a slice of the suite re-run in a child process under the electric-fence allocator (tests/efence/efence_alloc.cpp,
EOD_TEST_EFENCE=1): every device tensor of that process -- program buffers, packed weights, parameters, the tensors the kernel tests
hand to the C ABI -- is its own hipMalloc and ends where the allocation ends, new memory is NaN-filled.  An out-of-bounds access of any
launch is a memory fault there (the child dies, this test fails with its log), a read of never-written memory a NaN.  The whole
-m gpu suite is run this way by hand each round (`EOD_TEST_EFENCE=1 python -m pytest tests -m gpu`: 860 tests green in round 3);
this standing test keeps the cases that found the round-3 bug plus the kernels with the most intricate tails."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, timeout=900):
    env = dict(os.environ, EOD_TEST_EFENCE="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-m", "gpu", "-x", "-p", "no:cacheprovider", *args], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=timeout)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and " failed" not in r.stdout, tail


def test_unet_fuzz_cases_under_the_electric_fence():
    node = "tests/test_gpu_fuzz_archs.py::test_random_unet_configuration_vs_oracle"
    _run([f"{node}[{i}]" for i in (1, 8, 11, 17, 22)])   # (1 and 11: the per-sample bias read behind the last image, round 3)


def test_kernel_tests_under_the_electric_fence():
    _run(["tests/test_gpu_kernels.py", "-k",
          "conv_vs_torch or fused_1x1_skip or parity_class_form_vs_torch or attention_forward_natural_layout or first_conv or head_conv or gemm_nt"])


def test_training_fuzz_cases_under_the_electric_fence():
    node = "tests/test_gpu_fuzz_archs.py::test_random_unet_training_step_vs_oracle"
    _run([f"{node}[{i}-{p}]" for i in (1, 9, 12) for p in ("fp32", "fp16")])
