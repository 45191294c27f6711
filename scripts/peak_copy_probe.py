import sys, os
sys.path.insert(0, os.getcwd())
import torch
from reformer_tts_amd import _lib
dev=torch.device("cuda:0")
n=1<<30
src=torch.zeros(n,dtype=torch.uint8,device=dev); dst=torch.empty(n,dtype=torch.uint8,device=dev)
s=torch.cuda.current_stream().cuda_stream
a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
for _ in range(2): _lib.call("rtts_peak_copy",src.data_ptr(),dst.data_ptr(),n,s)
a.record()
for _ in range(5): _lib.call("rtts_peak_copy",src.data_ptr(),dst.data_ptr(),n,s)
b.record(); torch.cuda.synchronize()
print(os.environ.get("RTTS_PEAK_COPY_GRID"), "copy GB/s", 5*2*n/(a.elapsed_time(b)*1e-3)/1e9)
a.record()
for _ in range(5): dst.copy_(src)
b.record(); torch.cuda.synchronize()
print("torch copy_ GB/s", 5*2*n/(a.elapsed_time(b)*1e-3)/1e9)
