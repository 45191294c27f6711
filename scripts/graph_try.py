import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from reformer_tts_amd.model.config import baseline_model_config, baseline_training_config
from reformer_tts_amd.training import Trainer, build_model, synthetic_batch
from reformer_tts_amd.model.lsh_attention import LSHSelfAttention
dev = torch.device("cuda:0")
model = build_model(baseline_model_config(), dev)
tr = Trainer(model, baseline_training_config(), dev)
batch = synthetic_batch(12, 200, 1024, device=dev)
for m in model.modules():
    if isinstance(m, LSHSelfAttention):
        m._gen = None
        m.use_default_generator = True
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        tr.train_step(batch)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    tr.train_step(batch)
torch.cuda.synchronize()
print("eager ms/step", (time.perf_counter() - t0) * 100)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = tr.train_step(batch)
torch.cuda.synchronize()
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    g.replay()
torch.cuda.synchronize()
print("graph ms/step", (time.perf_counter() - t0) * 100, "loss", float(out[0]))
