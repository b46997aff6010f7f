"""Shared helpers of the parity tests and of __graft_entry__.smoke(): build the HIP-backed model from an
oracle/reference state-dict and compare against the oracle (the checker -- never the thing measured)."""
import torch

from oracle import wan_dit as W


def rel_rms(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).pow(2).mean().sqrt() / (b.pow(2).mean().sqrt() + 1e-12)).item()


def model_cfg(cfg):
    keys = ("patch_size", "num_attention_heads", "attention_head_dim", "in_channels", "out_channels", "text_dim",
            "freq_dim", "ffn_dim", "num_layers", "cross_attn_norm", "eps", "rope_max_seq_len")
    out = {k: cfg[k] for k in keys if k in cfg}
    out["patch_size"] = tuple(int(v) for v in out["patch_size"])
    out["cross_attn_norm"] = bool(out.get("cross_attn_norm", True))
    return out


def hip_wan_model(cfg, sd, device, dtype=torch.bfloat16):
    from frameino_amd.transformer_wan import WanTransformer3DModel
    m = WanTransformer3DModel(**model_cfg(cfg)).to(device)
    m.load_reference_state_dict(sd, dtype=dtype)
    return m.eval()


def bf16_state_dict(sd, dtype=torch.bfloat16):
    """The reference's bf16 deployment: everything in `dtype` except the fp32 islands (transformer_wan.py:393)."""
    return {k: (v.float() if any(s in k for s in W.FP32_KEEP) else v.to(dtype)) for k, v in sd.items()}


def smoke_check(device):
    """Tiny Wan DiT forward + one CFG/Euler step through the HIP path, against the oracle run on the CPU."""
    from frameino_amd import ops
    cfg = dict(W.WAN22_5B_CFG, num_attention_heads=2, attention_head_dim=128, in_channels=8, out_channels=4,
               text_dim=64, ffn_dim=512, num_layers=2)
    sd = W.wan_random_state_dict(cfg, seed=3, dtype=torch.float32, std=0.05)
    g = torch.Generator().manual_seed(4)
    x = torch.randn(1, 8, 3, 8, 12, generator=g)
    txt = torch.randn(1, 32, 64, generator=g)
    L = 3 * 4 * 6
    ts = torch.full((1, L), 700.0)
    ts[0, :24] = 0.0
    ref = W.wan_forward(sd, cfg, x, ts, txt)
    m = hip_wan_model(cfg, sd, device)
    out = m(x.to(device).bfloat16(), ts.to(device), txt.to(device).bfloat16(), return_dict=False)[0]
    torch.cuda.synchronize()
    r = rel_rms(out, ref)
    assert r < 3e-2, f"HIP Wan forward vs oracle rel-RMS {r}"
    # one sampler step
    lat = torch.randn(4, 3, 8, 12, generator=g)
    lat_d = lat.to(device)
    dt = torch.tensor([-0.05], device=device)
    ops.cfg_euler_step_(lat_d, out[0], out[0], 5.0, dt, round_out=False)
    exp = lat + (-0.05) * out[0].float().cpu()
    assert torch.allclose(lat_d.cpu(), exp, atol=1e-5)
    return r


def record(test, metric, value, bound, lower_is_better=True):
    """Append what a parity test MEASURED (not just that it passed) to gpurun_out/parity.jsonl -- one JSON object per
    line: {test, metric, value, bound, ok} -- so that the numbers behind `parity: green` are on file (the builder copies
    the file of the round's last full GPU run to profiles/rNN_parity.json).  Never raises: a read-only tree only loses
    the log."""
    import json
    import os
    ok = (value <= bound) if lower_is_better else (value >= bound)
    try:
        root = os.environ.get("FINO_PARITY_LOG_DIR") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                    "gpurun_out")
        os.makedirs(root, exist_ok=True)
        with open(os.path.join(root, "parity.jsonl"), "a") as f:
            f.write(json.dumps({"test": test, "metric": metric, "value": float(value), "bound": float(bound),
                                "ok": bool(ok)}) + "\n")
    except OSError:
        pass
    return ok
