"""Calibrate rocprofv3 FETCH_SIZE / WRITE_SIZE against known byte counts (torch copy kernels)."""
import torch
x = torch.empty(256 * 1024 * 1024 // 4, dtype=torch.float32, device="cuda").normal_()   # 256 MiB
y = torch.empty_like(x)
torch.cuda.synchronize()
for _ in range(3):
    y.copy_(x)            # reads 256 MiB, writes 256 MiB
torch.cuda.synchronize()
z = (x.view(-1, 96)[:, :32]).contiguous()   # strided 128-byte row pieces out of 384-byte rows
torch.cuda.synchronize()
