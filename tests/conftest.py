import gzip
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def vp():
    import vp_loader
    mod = vp_loader.load()
    mod.build()
    return mod


@pytest.fixture(scope="session")
def ob():
    import oracle_binding
    oracle_binding.lib()
    return oracle_binding


@pytest.fixture(scope="session")
def golden():
    return json.load(open(os.path.join(GOLDEN, "golden.json")))


@pytest.fixture(scope="session")
def pws_path(tmp_path_factory):
    """The reference's data/SHA256_64.pws, shipped gzip-compressed as a fixture."""
    p = tmp_path_factory.mktemp("pws") / "SHA256_64.pws"
    with gzip.open(os.path.join(GOLDEN, "SHA256_64.pws.gz"), "rb") as f:
        p.write_bytes(f.read())
    return str(p)


_GPU_COUNT = None


def gpu_count():
    """GPUs of this box, counted in a CHILD process: importing torch into a process that already holds libvpgpu.so (and with it /opt/rocm's HIP
    runtime) would load PyTorch's own bundled ROCm libraries beside them — two HIP runtimes, two rocm_smi copies whose same-named globals are
    destroyed twice at exit."""
    global _GPU_COUNT
    if _GPU_COUNT is None:
        import subprocess
        r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], stdout=subprocess.PIPE,
                           stderr=subprocess.DEVNULL, text=True, timeout=600)
        _GPU_COUNT = int(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 and r.stdout.strip() else 0
    return _GPU_COUNT


def gkr_slice(golden, name):
    g = golden[name]
    data = open(os.path.join(GOLDEN, g["transcript"]), "rb").read()
    return data[g["gkr_slice"][0]: g["gkr_slice"][1]]


@pytest.fixture(scope="session")
def gold_gkr(golden):
    return lambda name: gkr_slice(golden, name)
