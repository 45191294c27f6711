import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from reformer_tts_amd.model.config import baseline_model_config, baseline_training_config
from reformer_tts_amd.training import Trainer, build_model, synthetic_batch
dev = torch.device("cuda:0")
zero_drop = "--nodrop" in sys.argv
model = build_model(baseline_model_config(), dev)
if zero_drop:
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout): m.p = 0.0
tr = Trainer(model, baseline_training_config(), dev)
batch = synthetic_batch(12, 200, 1024, device=dev)
tr.capture(batch)
for i in range(4):
    out = tr.replay()
    torch.cuda.synchronize()
    print("replay", i, [float(x) for x in out], "param nan:", bool(torch.isnan(tr.flat_p).any()), "grad nan:", bool(torch.isnan(tr.flat_g).any()), flush=True)
    if torch.isnan(tr.flat_g).any():
        bad = [n for n, (s, e) in tr.offsets.items() if torch.isnan(tr.flat_g[s:e]).any()]
        print(" nan grads in", bad[:8], len(bad))
        break
for i in range(12):
    out = tr.replay()
torch.cuda.synchronize()
print("after 16 replays", [float(x) for x in out], bool(torch.isnan(tr.flat_p).any()))
loss = out[0]
for i in range(3):
    o2 = tr.train_step(batch)
    torch.cuda.synchronize()
    print("eager after graph", i, [float(x) for x in o2], "static loss now:", float(loss), "param nan:", bool(torch.isnan(tr.flat_p).any()))
