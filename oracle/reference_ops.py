"""Eager-PyTorch execution of the projector with the SAME torch op sequence the reference module
issues (nn.Linear / nn.GELU / nn.LayerNorm / F.interpolate / nn.MultiheadAttention with L=1, S=s*s,
token-major layout; reference builder.py:107-137), driven off the parameter containers of
``tokenpacker_amd.TokenPacker``.  TEST INFRASTRUCTURE (only tests/, tools/ and the cpu_baseline leg of bench.py import it): it provides (a) the "reference on
PyTorch-ROCm eager" baseline the >=5x target of BASELINE.json is quoted against, and (b) an
independent bf16 sanity reference on the GPU.  The product never imports it."""
import torch
import torch.nn.functional as F


def regions_token_major(t, side, k):
    """[side*side, B, c] -> [k*k, (side/k)^2 * B, c]; entry [a*k+b, (i*(side/k)+j)*B + n] is token
    (i*k+a, j*k+b) of image n  (what the reference's divide_feature produces, builder.py:96-105)."""
    n_tok, B, c = t.shape
    G = side // k
    t = t.reshape(G, k, G, k, B, c)            # (i, a, j, b, n, c)
    t = t.permute(1, 3, 0, 2, 4, 5)            # (a, b, i, j, n, c)
    return t.reshape(k * k, G * G * B, c)


def eager_forward(m, x, x_multi):
    key = m.ln_k_1(m.k_proj_1(x_multi)).permute(1, 0, 2)
    value = m.ln_v_1(m.v_proj_1(x_multi)).permute(1, 0, 2)
    n_tok, B, c = key.shape
    g, G, s = m.raw_grid, m.grid_size, m.scale_factor
    q = F.interpolate(x.reshape(B, g, g, -1).float().permute(0, 3, 1, 2), size=(G, G), mode="bilinear")
    q = q.permute(0, 2, 3, 1).reshape(B, G * G, -1).to(x.dtype)
    query = m.ln_q_1(m.q_proj_1(q)).permute(1, 0, 2)
    out = m.clip_attn(regions_token_major(query, G, 1), regions_token_major(key, g, s),
                      regions_token_major(value, g, s), attn_mask=None)[0]
    out = out.reshape(m.num_queries, B, -1).permute(1, 0, 2)
    return m.mlp(out)
