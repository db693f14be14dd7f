"""Debug aid: chained forward on GPU vs oracle (bf16 mode and fp32 mode), per-layer full-tensor and CLS-row errors."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
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
outs = []
hooks = [blk.register_forward_hook(lambda mod, i, o: outs.append(o.cpu())) for blk in m.encoder.layers]
with torch.no_grad():
    logits = m(x.cuda()).cpu()
tb = O.embed_tokens(x, sd, cfg, "bf16") + O._t(sd, "encoder.pos_embedding")
tf = O.embed_tokens(x, sd, cfg, "fp32") + O._t(sd, "encoder.pos_embedding")
for li in range(cfg["num_layers"]):
    tb = O.vit_block(tb, sd, f"encoder.layers.{li}.", cfg["num_heads"], 1e-5, "bf16")
    tf = O.vit_block(tf, sd, f"encoder.layers.{li}.", cfg["num_heads"], 1e-5, "fp32")
    print(f"L{li}: hip-vs-bf16oracle all {rel(outs[li], tb):.2e} cls {rel(outs[li][:,0], tb[:,0]):.2e} | hip-vs-fp32 all {rel(outs[li], tf):.2e} cls {rel(outs[li][:,0], tf[:,0]):.2e} | bf16oracle-vs-fp32 all {rel(tb, tf):.2e}")
lb = O.vit_forward(x, sd, cfg, "bf16"); lf = O.vit_forward(x, sd, cfg, "fp32")
print("logits hip-vs-bf16oracle", rel(logits, lb), "hip-vs-fp32", rel(logits, lf), "bf16oracle-vs-fp32", rel(lb, lf))
with torch.no_grad():
    tok = engine.embed_tokens(m, x.cuda()).cpu()
t0 = O.embed_tokens(x, sd, cfg, "bf16") + O._t(sd, "encoder.pos_embedding")
print("tokens all", rel(tok, t0), "cls", rel(tok[:, 0], t0[:, 0]), "row1", rel(tok[:, 1], t0[:, 1]), "last", rel(tok[:, -1], t0[:, -1]))
with torch.no_grad():
    o_h = m.encoder.layers[0](tok.cuda()).cpu()
    o_o = m.encoder.layers[0](t0.cuda()).cpu()
ref = O.vit_block(t0, sd, "encoder.layers.0.", cfg["num_heads"], 1e-5, "bf16")
print("block0(hip tokens) cls", rel(o_h[:, 0], ref[:, 0]), " block0(oracle tokens) cls", rel(o_o[:, 0], ref[:, 0]))
print("max abs token diff", (tok - t0).abs().max().item(), "at", (tok - t0).abs().flatten().argmax().item())
