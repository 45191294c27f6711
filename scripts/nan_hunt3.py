import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
_empty, _empty_like = torch.empty, torch.empty_like
def poison(t):
    if t.is_cuda:
        if t.dtype.is_floating_point: t.fill_(float("nan"))
        elif t.dtype in (torch.int32, torch.int64): t.fill_(-123456789)
    return t
torch.empty = lambda *a, **k: poison(_empty(*a, **k))
torch.empty_like = lambda *a, **k: poison(_empty_like(*a, **k))
from reformer_tts_amd.model.config import baseline_model_config, baseline_training_config
from reformer_tts_amd.training import Trainer, build_model, synthetic_batch
dev = torch.device("cuda:0")
model = build_model(baseline_model_config(), dev)
for m in model.modules():
    if isinstance(m, torch.nn.Dropout): m.p = 0.0
tr = Trainer(model, baseline_training_config(), dev)
batch = synthetic_batch(12, 200, 1024, device=dev)
for i in range(2):
    out = tr.train_step(batch); torch.cuda.synchronize()
    bad = [n for n, (s, e) in tr.offsets.items() if torch.isnan(tr.flat_g[s:e]).any()]
    print(i, [float(x) for x in out], "nan grads:", len(bad), bad[:10], flush=True)
