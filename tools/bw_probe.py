import torch, time
x = torch.empty(2**29, dtype=torch.float64, device='cuda').normal_()   # 4 GiB
for _ in range(3): x.sum()
torch.cuda.synchronize(); t=time.perf_counter()
for _ in range(10): x.sum()
torch.cuda.synchronize(); dt=(time.perf_counter()-t)/10
print('torch sum read GB/s', x.numel()*8/dt/1e9)
y = torch.empty_like(x)
for _ in range(3): y.copy_(x)
torch.cuda.synchronize(); t=time.perf_counter()
for _ in range(10): y.copy_(x)
torch.cuda.synchronize(); dt=(time.perf_counter()-t)/10
print('torch copy r+w GB/s', 2*x.numel()*8/dt/1e9)
x = torch.empty(2**24, dtype=torch.float64, device='cuda').normal_()   # 128 MiB: Infinity-Cache sized
for _ in range(5): x.sum()
torch.cuda.synchronize(); t=time.perf_counter()
for _ in range(50): x.sum()
torch.cuda.synchronize(); dt=(time.perf_counter()-t)/50
print('torch sum 128MiB (MALL?) GB/s', x.numel()*8/dt/1e9)
