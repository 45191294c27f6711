import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from reformer_tts_amd.model.config import baseline_model_config, baseline_training_config
from reformer_tts_amd.training import Trainer, build_model, synthetic_batch
dev = torch.device("cuda:0")
mode = sys.argv[1]
model = build_model(baseline_model_config(), dev)
for m in model.modules():
    if isinstance(m, torch.nn.Dropout): m.p = 0.0
tr = Trainer(model, baseline_training_config(), dev)
batch = synthetic_batch(12, 200, 1024, device=dev)
if mode == "graph":
    tr.capture(batch); step = tr.replay
else:
    for _ in range(2): tr.train_step(batch)
    step = lambda: tr.train_step(batch)
prev = tr.flat_p.clone()
for i in range(6):
    out = step(); torch.cuda.synchronize()
    d = (tr.flat_p - prev).norm().item(); prev = tr.flat_p.clone()
    print(mode, i, "loss %.4f" % float(out[0]), "hyper", tr.hyper.tolist(), "scale", tr.ws_scale.tolist(), "dp %.5f" % d, "gstep", tr.global_step, flush=True)
