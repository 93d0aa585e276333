"""Developer probe (GPU box): does PyTorch's caching allocator on this build ever hand a block freed under stream A to an allocation under
stream B (without a synchronisation)?  Pure PyTorch, no library kernel."""
import torch
dev = "cuda:0"
N = 32 * 1024 * 1024
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
print("stream ids:", sa.cuda_stream, sb.cuda_stream, "default:", torch.cuda.current_stream().cuda_stream)
shared, bad = 0, 0
for it in range(200):
    with torch.cuda.stream(sa):
        x = torch.empty(N, device=dev)
        px = x.data_ptr()
        x.fill_(1.0)
        for _ in range(6): x.mul_(1.0)            # keep stream A busy on x for a while
        y = x * 2.0
        del x
    with torch.cuda.stream(sb):
        z = torch.empty(N, device=dev)
        shared += int(z.data_ptr() == px)
        z.fill_(7.0)
    sa.synchronize(); sb.synchronize()
    bad += int(float(y.max()) != 2.0 or float(y.min()) != 2.0)
    del y, z
print(f"block freed under stream A handed to stream B: {shared} of 200 iterations; wrong results: {bad}")
