"""The driver's entry points in a fresh interpreter: smoke() must work when libslow5gpu.so is the first thing that touches
HIP in the process (the torch wheel ships its own HIP runtime; slow5tools_amd/_lib.py orders the loads)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_smoke_in_a_fresh_process():
    r = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.smoke()"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "smoke ok" in r.stdout


@pytest.mark.gpu
def test_host_batch_call_before_any_torch_use():
    code = ("import numpy as np\n"
            "from slow5tools_amd import press\n"
            "r = press.encode_records([np.arange(1000, dtype=np.int16)], [press.pack_hdr(b'x', 0, 1.0, 2.0, 3.0, 4.0)])\n"
            "d = press.decode_records([r[0][8:]])\n"
            "assert d[0]['status'] == 0 and (d[0]['signal'] == np.arange(1000)).all()\n"
            "import torch\n"
            "assert torch.cuda.is_available() and torch.zeros(4, device='cuda:0').sum().item() == 0\n"
            "print('ok')\n")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr
