"""Drop-in boundary of the nn.Module, checked without a GPU."""
import glob
import os

import numpy as np
import pytest
import torch

from tokenpacker_amd import TokenPacker, build_vision_projector, synth


def test_state_dict_contract_names_shapes_order():
    for D in (4096, 5120, 256):
        m = TokenPacker(hidden_size=D)
        sd = m.state_dict()
        want = synth.param_shapes(D)
        assert list(sd.keys()) == list(want.keys())
        for k, shape in want.items():
            assert tuple(sd[k].shape) == shape, k
    assert sum(p.numel() for p in TokenPacker(hidden_size=4096).parameters()) == 36_722_688


def test_load_reference_style_state_dict_strict():
    params = synth.make_params(3, 256)
    m = TokenPacker(hidden_size=256, scale_factor=3)
    res = m.load_state_dict(params, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    for k, v in m.state_dict().items():
        assert torch.equal(v, params[k])
    # projector-only checkpoints are stored with an 'mm_projector.' prefix and loaded through
    # get_w() (llava_arch.py:80-83): emulate it
    ckpt = {"model.mm_projector." + k: v for k, v in params.items()}
    stripped = {k.split("mm_projector.")[1]: v for k, v in ckpt.items() if "mm_projector" in k}
    TokenPacker(hidden_size=256).load_state_dict(stripped)


def test_constructor_and_factory_mirror_reference():
    cfg = type("Cfg", (), {"hidden_size": 5120, "scale_factor": 4, "mm_projector_type": "tokenpacker"})()
    m = build_vision_projector(cfg)
    assert isinstance(m, TokenPacker)
    assert (m.raw_grid, m.grid_size, m.num_queries, m.scale_factor, m.embed_dim, m.num_heads) == (24, 6, 36, 4, 1024, 8)
    assert m.mlp[2].weight.shape == (5120, 5120)
    with pytest.raises(ValueError, match="scale_factor must be divisible by grid size"):
        TokenPacker(scale_factor=5)
    # default init follows the reference: zero biases, unit LayerNorm, small linear weights
    m = TokenPacker(hidden_size=256)
    assert float(m.k_proj_1[0].bias.abs().max()) == 0.0 and float(m.ln_k_1.weight.min()) == 1.0
    assert 0.015 < float(m.q_proj_1.weight.std()) < 0.025
    assert all(p.requires_grad for p in m.parameters())
    m.requires_grad_(False)
    assert not any(p.requires_grad for p in m.parameters())


def test_cpu_tensors_raise_instead_of_falling_back():
    m = TokenPacker(hidden_size=256).requires_grad_(False)
    x = torch.zeros(1, 576, 1024, dtype=torch.bfloat16)
    xm = torch.zeros(1, 576, 4096, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m((x, xm))


def test_product_never_imports_the_oracle():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for path in glob.glob(os.path.join(root, "tokenpacker_amd", "**", "*.py"), recursive=True):
        src = open(path).read()
        assert "import oracle" not in src and "from oracle" not in src, path
    bench = open(os.path.join(root, "bench.py")).read()
    # bench.py may use the oracle only inside its cpu_baseline leg
    assert bench.count("from oracle") + bench.count("import oracle") <= 1


def test_golden_param_digests_are_reproducible():
    for path in glob.glob(os.path.join(os.path.dirname(__file__), "golden", "s[0-9]_D*.npz")):
        z = np.load(path)
        p = synth.make_params(int(z["param_seed"]), int(z["hidden_size"]))
        assert synth.tensor_digest(*p.values()) == str(z["params_sha256"])


def test_deepcopy_and_pickle_drop_the_kernel_side_caches():
    """The reference module can be deep-copied and pickled (HF Trainer, EMA copies, torch.save(model)); ours carries
    device caches that must not travel (a HIP event is not picklable; a copied weight image would alias the original)."""
    import copy
    import pickle
    m = TokenPacker(hidden_size=256, scale_factor=3)
    m._packed = torch.zeros(4)                           # stand-ins for what a GPU forward leaves behind
    m._packed_key = ("k",)
    m._packed_event = (i for i in range(1))              # a generator: neither picklable nor deep-copyable, like a HIP event
    m._workspaces[("dev", 0)] = torch.zeros(8)
    m._last_launch = (object(), torch.zeros(1), 0)
    for clone in (copy.deepcopy(m), pickle.loads(pickle.dumps(m))):
        assert clone._packed is None and clone._packed_key is None and clone._packed_event is None
        assert clone._workspaces == {} and clone._last_launch is None
        assert clone.scale_factor == 3 and clone.hidden_size == 256
        for (k, a), (_, b) in zip(m.state_dict().items(), clone.state_dict().items()):
            assert torch.equal(a, b) and a.data_ptr() != b.data_ptr(), k
    assert m._packed is not None                         # the original keeps its caches


def test_partitioned_parameters_are_refused_with_a_message():
    """DeepSpeed ZeRO-3 leaves empty placeholders where the parameters were; this module never calls its child containers, so the
    per-submodule gather hooks never fire — the pack must say so instead of handing empty tensors to the kernels."""
    import pytest
    import torch
    from tokenpacker_amd import TokenPacker
    m = TokenPacker(hidden_size=256)
    m.mlp[2].weight.data = torch.empty(0)                    # what a partitioned parameter looks like outside its gather context
    with pytest.raises(RuntimeError, match="ZeRO-3"):
        m._ensure_packed(torch.float32, torch.device("cpu"), 0)


def test_lib_variant_name_is_validated(monkeypatch):
    import pytest
    from tokenpacker_amd import _capi
    monkeypatch.setenv("TP_LIB_VARIANT", "../evil")
    with pytest.raises(ValueError):
        _capi._lib_name()
    monkeypatch.setenv("TP_LIB_VARIANT", "s1")
    with pytest.warns(RuntimeWarning):
        assert _capi._lib_name() == "libtokenpacker_s1.so"
    monkeypatch.delenv("TP_LIB_VARIANT")
    assert _capi._lib_name() == "libtokenpacker_hip.so"
