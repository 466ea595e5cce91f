"""The C ABI driven from plain C (examples/c_abi_demo.c: hipMalloc / tp_pack_weights / tp_forward, no Python, no torch),
compiled with gcc (a C11 host program: the HIP runtime API and the library are all it links) on the GPU box and compared bit for bit with the nn.Module on the same bytes — the drop-in boundary is
the shared library, the Python module is one caller of it."""
import os
import shutil
import subprocess

import pytest
import torch

from tokenpacker_amd import TokenPacker, _capi, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("s,dtype", [(2, torch.bfloat16), (3, torch.float16)])
def test_c_caller_matches_the_module_bit_for_bit(tmp_path, s, dtype):
    gcc = shutil.which("gcc") or "gcc"
    exe = str(tmp_path / "tp_demo")
    build = subprocess.run([gcc, "-O2", "-std=c11", os.path.join(ROOT, "examples", "c_abi_demo.c"),
                            "-I" + os.path.join(ROOT, "include"), "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__",
                            "-L" + os.path.join(ROOT, "tokenpacker_amd"), "-ltokenpacker_hip", "-L/opt/rocm/lib", "-lamdhip64",
                            "-Wl,-rpath," + os.path.join(ROOT, "tokenpacker_amd"), "-Wl,-rpath,/opt/rocm/lib", "-o", exe],
                           capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-3000:]
    B, D = 3, 256
    params = synth.make_params(33, D)
    x, xm = synth.make_inputs(34, B, dtype)
    d = tmp_path / "io"
    d.mkdir()
    for name, t in params.items():
        t.to(dtype).contiguous().view(torch.int16).numpy().tofile(str(d / (name.replace(".", "_") + ".bin")))
    x.contiguous().view(torch.int16).numpy().tofile(str(d / "x.bin"))
    xm.contiguous().view(torch.int16).numpy().tofile(str(d / "x_multi.bin"))
    run = subprocess.run([exe, str(d), str(B), str(s), str(D), str(_capi.TP_BF16 if dtype == torch.bfloat16 else _capi.TP_F16)],
                         capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "deterministic, saturated=0" in run.stdout
    M = (24 // s) ** 2
    import numpy as np
    y_c = torch.from_numpy(np.fromfile(str(d / "out.bin"), dtype=np.int16)).view(dtype).reshape(B, M, D)
    m = TokenPacker(hidden_size=D, scale_factor=s)
    m.load_state_dict(params)
    m = m.to(device="cuda", dtype=dtype).eval().requires_grad_(False)
    with torch.no_grad():
        y = m((x.cuda(), xm.cuda())).cpu()
    assert torch.equal(y_c, y)
