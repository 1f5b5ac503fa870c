"""GPU tool: pinned host -> device copy bandwidth, one stream vs several concurrent streams / chunk sizes."""
import time, torch
N = 480 * 1024 * 1024
host = [torch.empty(N, dtype=torch.uint8).pin_memory() for _ in range(2)]
dev = [torch.empty(N, dtype=torch.uint8, device="cuda") for _ in range(2)]
def run(nstreams, chunks):
    sts = [torch.cuda.Stream() for _ in range(nstreams)]
    cs = N // chunks
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for rep in range(4):
        for c in range(chunks):
            with torch.cuda.stream(sts[c % nstreams]):
                dev[rep % 2][c * cs:(c + 1) * cs].copy_(host[rep % 2][c * cs:(c + 1) * cs], non_blocking=True)
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    return 4 * N / el / 1e9
for ns, ch in ((1, 1), (1, 8), (2, 2), (2, 8), (4, 4), (4, 16), (8, 8)):
    run(ns, ch)
    print(f"streams {ns} chunks {ch}: {run(ns, ch):6.1f} GB/s", flush=True)
d2 = torch.empty(N, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(4): d2.copy_(dev[0])
torch.cuda.synchronize(); print(f"d2d: {4*N/(time.perf_counter()-t0)/1e9:6.1f} GB/s")
