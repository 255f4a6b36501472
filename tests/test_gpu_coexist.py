"""The library and a host framework share ONE HIP runtime whichever is loaded first: a process that
creates a handle through libcadnip_hip.so first and imports torch afterwards must still see the GPU from torch -- and the
other way round.  Each order runs in its own interpreter."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

LIB_FIRST = """
import sys; sys.path.insert(0, %r)
import numpy as np
import cadnip_jl_amd as cj
from cadnip_jl_amd import hip
from tests import circuits as tc
circ = tc.divider()
st = cj.discover(circ, {})
h = hip.Handle(st, 2)
h.set_params(cj.pack_params(st, circ, {}, np.full(2, 27.0), 2))
h.rebuild(np.zeros((2, st.n)), 0.0)
import torch
assert torch.cuda.is_available(), "torch sees no device after the library initialised HIP"
x = torch.arange(8, device="cuda", dtype=torch.float64)
assert float(x.sum().item()) == 28.0
h.rebuild(np.ones((2, st.n)), 0.0)          # and the library still works next to torch's context
G, C, b, _ = h.get_GCb()
assert np.all(np.isfinite(G))
h.close()
print("COEXIST_OK lib-first")
"""

TORCH_FIRST = """
import sys; sys.path.insert(0, %r)
import torch
assert torch.cuda.is_available()
x = torch.ones(4, device="cuda")
import numpy as np
import cadnip_jl_amd as cj
from cadnip_jl_amd import hip
from tests import circuits as tc
circ = tc.divider()
st = cj.discover(circ, {})
h = hip.Handle(st, 1)
h.set_params(cj.pack_params(st, circ, {}, np.full(1, 27.0), 1))
h.rebuild(np.zeros((1, st.n)), 0.0)
h.close()
assert float(x.sum().item()) == 4.0
print("COEXIST_OK torch-first")
"""


@pytest.mark.parametrize("script,tag", [(LIB_FIRST, "lib-first"), (TORCH_FIRST, "torch-first")])
def test_library_and_torch_share_the_gpu_in_either_load_order(script, tag):
    p = subprocess.run([sys.executable, "-c", script % ROOT], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert p.returncode == 0 and ("COEXIST_OK " + tag) in p.stdout, p.stdout[-2000:]
