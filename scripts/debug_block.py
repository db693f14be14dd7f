"""Debug aid: run ONE encoder block on the GPU and compare every intermediate with the oracle (bf16 mode)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import torch.nn.functional as F
from oracle import vit_oracle as O
from peekvit_amd import synth, engine
from peekvit_amd.models.vit import VisionTransformer

def rel(a, b):
    a = a.double().cpu(); b = b.double().cpu()
    return float((a - b).norm() / b.norm())

name = sys.argv[1] if len(sys.argv) > 1 else "vit_tiny"
cfg = synth.MODEL_CONFIGS[name]
sd = synth.synth_state_dict(cfg)
m = VisionTransformer(**cfg); synth.load_synth_weights(m, cfg); m = m.eval().cuda()
x = torch.from_numpy(synth.synth_images(2, cfg["image_size"]))
H = cfg["num_heads"]; D = cfg["hidden_dim"]; dh = D // H
with torch.no_grad():
    tok = engine.embed_tokens(m, x.cuda())
    t_ref = O.embed_tokens(x, sd, cfg, "bf16") + O._t(sd, "encoder.pos_embedding")
    print("tokens", rel(tok, t_ref))
    B, S, _ = tok.shape
    # feed the ORACLE tokens to the HIP block so errors do not accumulate
    cur = t_ref.clone()
    for li in range(cfg["num_layers"]):
        p = f"encoder.layers.{li}."
        g = lambda k: O._t(sd, p + k)
        out = m.encoder.layers[li](cur.cuda())
        ws = engine.workspace
        qkv = ws.get("qkv", (B, S, 3 * D), torch.bfloat16, tok.device).float().cpu()
        att = ws.get("att", (B, S, D), torch.bfloat16, tok.device).float().cpu()
        x1 = ws.get("x1", (B, S, D), torch.float32, tok.device).cpu()
        gg = ws.get("g", (B, S, cfg["mlp_dim"]), torch.bfloat16, tok.device).float().cpu()
        h2 = ws.get("h", (B, S, D), torch.bfloat16, tok.device).float().cpu()
        # oracle intermediates
        h = O.layer_norm(cur, g("ln_1.weight"), g("ln_1.bias"), 1e-5)
        r_qkv = O.linear(h, g("self_attention.self_attention.in_proj_weight"), g("self_attention.self_attention.in_proj_bias"), "bf16")
        r_qkv[..., :D] *= dh ** -0.5
        r_qkv = O.rb(r_qkv, "bf16")
        q, k, v = (t.reshape(B, S, H, dh).transpose(1, 2) for t in r_qkv.split(D, dim=-1))
        r_att = O.rb(O.attention_core(q, k, v, "bf16").transpose(1, 2).reshape(B, S, D), "bf16")
        r_x1 = cur + O.linear(r_att, g("self_attention.self_attention.out_proj.weight"), g("self_attention.self_attention.out_proj.bias"), "bf16")
        r_h2 = O.rb(O.layer_norm(r_x1, g("ln_2.weight"), g("ln_2.bias"), 1e-5), "bf16")
        r_g = O.rb(F.gelu(O.linear(r_h2, g("mlp.fc1.weight"), g("mlp.fc1.bias"), "bf16")), "bf16")
        r_out = r_x1 + O.linear(r_g, g("mlp.fc2.weight"), g("mlp.fc2.bias"), "bf16")
        print(f"L{li}: qkv {rel(qkv, r_qkv):.2e} att {rel(att, r_att):.2e} x1 {rel(x1, r_x1):.2e} h2 {rel(h2, r_h2):.2e} "
              f"g {rel(gg, r_g):.2e} out {rel(out, r_out):.2e}   (mismatched qkv elems {(qkv != r_qkv).float().mean():.2e}, att {(att != r_att).float().mean():.2e})")
        r0 = lambda a, b: rel(a.reshape(B, S, -1)[:, 0], b.reshape(B, S, -1)[:, 0])
        hip_h1 = None
        print(f"     row0: qkv {r0(qkv, r_qkv):.2e} att {r0(att, r_att):.2e} x1 {r0(x1, r_x1):.2e} h2 {r0(h2, r_h2):.2e} g {r0(gg, r_g):.2e} out {r0(out.cpu(), r_out):.2e}")
        # which rows of att are worst?
        e = (att - r_att).reshape(B, S, -1).norm(dim=-1) / r_att.reshape(B, S, -1).norm(dim=-1)
        print("     worst att rows", torch.topk(e.flatten(), 5))
        e = (qkv - r_qkv).reshape(B, S, -1).norm(dim=-1) / r_qkv.reshape(B, S, -1).norm(dim=-1)
        print("     worst qkv rows", torch.topk(e.flatten(), 5))
        cur = r_out
