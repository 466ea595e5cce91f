"""x_multi consumed as the tower's four hidden-state slices (tp_forward_parts / tokenpacker_amd.tower) must be
bit-identical to the concatenated form the reference builds with torch.cat (clip_encoder.py:28-44) — forward,
training forward and every parameter gradient, at small and persistent-kernel batch sizes."""
import pytest
import torch

from tokenpacker_amd import TokenPacker, synth, tower

pytestmark = pytest.mark.gpu


def _hidden_states(B, dtype, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return [torch.randn(B, 577, 1024, generator=g, device="cuda").to(dtype) for _ in range(25)]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,s", [(2, 2), (3, 3), (40, 2)])       # B = 40: the first GEMM runs on the persistent kernel
def test_parts_equal_concatenated(dtype, B, s):
    hs = _hidden_states(B, dtype, 11)
    x, parts = tower.select_features(hs)
    x_ref, xm_ref = tower.concat_reference(hs)
    assert x.data_ptr() == parts[3].data_ptr()                    # select_layer -2 of 25 states IS hidden state 23
    assert not parts[0].is_contiguous() and torch.equal(torch.cat(parts, dim=2), xm_ref)
    m = TokenPacker(hidden_size=256, scale_factor=s)
    m.load_state_dict(synth.make_params(12, 256))
    m = m.to(device="cuda", dtype=dtype)
    with torch.no_grad():
        y_parts = m((x, parts))
        y_cat = m((x_ref, xm_ref))
    assert torch.equal(y_parts, y_cat)

    grads = []
    for xm in (parts, xm_ref):
        m.zero_grad(set_to_none=True)
        y = m((x, xm))
        y.float().square().sum().backward()
        grads.append((y.detach(), [p.grad.clone() for p in m.parameters()]))
    assert torch.equal(grads[0][0], grads[1][0])
    # inference runs the fused LayerNorm chain (and, for scale_factor >= 3, the absorbed K/V schedule); the training forward
    # cannot (the backward needs H2, K, V): the same function up to rounding — and bit-identical once inference is told to
    # run the training forward's schedule
    from tests import gpu_util as gu
    assert float((grads[0][0].float() - y_cat.float()).abs().max()) <= 2.0 ** -6 * float(y_cat.float().abs().max())
    with gu.training_schedule_for_inference():
        with torch.no_grad():
            assert torch.equal(m((x, parts)), grads[0][0])
    assert all(torch.equal(a, b) for a, b in zip(grads[0][1], grads[1][1]))


def test_parts_argument_errors():
    m = TokenPacker(hidden_size=256).to(device="cuda", dtype=torch.bfloat16).requires_grad_(False)
    hs = _hidden_states(1, torch.bfloat16, 1)
    x, parts = tower.select_features(hs)
    with pytest.raises(ValueError):
        m((x, parts[:3]))
    with pytest.raises(ValueError):
        m((x, (parts[0], parts[1], parts[2], parts[3][:, :, :512])))
    with pytest.raises(ValueError):
        tower.select_features(hs, select_feature="bogus")
