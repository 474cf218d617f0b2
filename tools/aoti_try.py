"""AOTInductor round trip of the exported whole-step program on the GPU box (what nequip-compile --mode aotinductor
produces and pair_allegro loads).  Prints max|dF| between the packaged model and the direct call."""
import os
import sys
import tempfile

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from allegro_amd.export import ExportableAllegro  # noqa: E402
from tests.golden_utils import load_model_fixture  # noqa: E402
from tests.hip_utils import fixture_data, model_from_fixture  # noqa: E402

dev = torch.device("cuda:0")
fx = load_model_fixture("c2", torch.float32)
m = model_from_fixture(fx, torch.float32, device=dev)
data, sv = fixture_data(fx, torch.float32, dev)
ex = ExportableAllegro(m, dev)
args = (data["pos"], data["edge_index"], data["atom_types"], sv)
want = ex(*args)
ep = torch.export.export(ex, args)
path = os.path.join(tempfile.mkdtemp(), "allegro_mi355x.pt2")
torch._inductor.aoti_compile_and_package(ep, package_path=path)
print("packaged:", os.path.getsize(path), "bytes")
runner = torch._inductor.aoti_load_package(path)
got = runner(*args)
print("AOTI max|dE_i| =", float((got[0] - want[0]).abs().max()), " max|dF| =", float((got[2] - want[2]).abs().max()),
      " vs golden max|dF| =", float((got[2].cpu() - fx["out"]["forces"]).abs().max()))
