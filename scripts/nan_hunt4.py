import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from reformer_tts_amd.model.config import baseline_model_config, baseline_training_config
from reformer_tts_amd.model.lsh_attention import LSHSelfAttention
from reformer_tts_amd.training import Trainer, build_model, synthetic_batch
dev = torch.device("cuda:0")
model = build_model(baseline_model_config(), dev)
for m in model.modules():
    if isinstance(m, torch.nn.Dropout): m.p = 0.0
    if isinstance(m, LSHSelfAttention):
        t = 1024 if m.causal else 256
        m.forced_rotations = torch.randn(1, 64, 8, t // m.bucket_size // 2, generator=torch.Generator().manual_seed(1)).to(dev)
tr = Trainer(model, baseline_training_config(), dev)
batch = synthetic_batch(12, 200, 1024, device=dev)
def fb():
    model.train(); tr.zero_grad()
    l = tr.forward_loss(batch); l[0].backward(); return l[0].detach()
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2): fb()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
ref = tr.flat_g.clone()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = fb()
for i in range(4):
    g.replay(); torch.cuda.synchronize()
    bad = [n for n, (s, e) in tr.offsets.items() if not torch.isfinite(tr.flat_g[s:e]).all()]
    diff = [(n, ((tr.flat_g[s:e]-ref[s:e]).norm()/(ref[s:e].norm()+1e-12)).item()) for n,(s,e) in tr.offsets.items()]
    diff.sort(key=lambda x:-x[1] if x[1]==x[1] else -1e9)
    print("replay", i, float(out), "nonfinite:", bad[:5], "max rel diff vs eager:", diff[:3], flush=True)
