import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950) GPU; run with -m gpu on the GPU box")


@pytest.fixture(scope="session")
def product():
    """The built product library (no GPU needed to load it)."""
    import qnnpack_amd
    if not os.path.exists(qnnpack_amd.library_path()):
        qnnpack_amd.build()
    return qnnpack_amd.load()


@pytest.fixture(scope="session")
def debug_hooks(product):
    """libqnnpack_gfx950_dbg.so: the create-time host logic exported for the CPU tier (not part of the product)."""
    import qnnpack_amd
    return qnnpack_amd.load_debug()


@pytest.fixture(scope="session")
def qnnp(product):
    """Product library bound to the GPU. Fails (never skips) when the device is unusable:
    a GPU-tier test passing without the HIP path would be a false parity claim."""
    import torch  # device memory + streams only
    assert torch.cuda.is_available(), "gpu-marked test running without a GPU"
    torch.cuda.set_device(0)
    torch.zeros(1, device="cuda")  # create the HIP context torch and the library share
    product.initialize()
    info = product.device_info()
    assert info["arch"].startswith("gfx950"), info
    product.set_stream(torch.cuda.current_stream().cuda_stream)
    product.set_async(False)
    product.set_option("gemm_kernel", 0)
    product.set_option("dwconv_kernel", 0)
    return product
