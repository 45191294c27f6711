import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from reformer_tts_amd.model.config import baseline_model_config, baseline_training_config
from reformer_tts_amd.training import Trainer, build_model, synthetic_batch
dev = torch.device("cuda:0")
mode = sys.argv[1] if len(sys.argv) > 1 else "graph"
model = build_model(baseline_model_config(), dev)
tr = Trainer(model, baseline_training_config(), dev)
batch = synthetic_batch(12, 200, 1024, device=dev)
if mode == "graph":
    tr.capture(batch)
    step = tr.replay
else:
    step = lambda: tr.train_step(batch)
for i in range(40):
    out = step()
    torch.cuda.synchronize()
    l = float(out[0])
    bad_p = bool(torch.isnan(tr.flat_p).any()); bad_g = bool(torch.isnan(tr.flat_g).any())
    if l != l or bad_p or bad_g or i % 10 == 0:
        print(mode, "step", i, [float(x) for x in out], "p nan", bad_p, "g nan", bad_g, flush=True)
    if l != l or bad_p or bad_g:
        bad = [n for n, (s, e) in tr.offsets.items() if torch.isnan(tr.flat_g[s:e]).any()]
        print(" nan grads:", len(bad), bad[:6])
        badp = [n for n, (s, e) in tr.offsets.items() if torch.isnan(tr.flat_p[s:e]).any()]
        print(" nan params:", len(badp), badp[:6])
        break
