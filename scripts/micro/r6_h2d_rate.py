"""Host -> device copy rate of the box: pinned memory, one stream and four streams, 1 / 4 / 64 MB pieces (what bounds the creates of many tables)."""
import time, torch
dev = torch.device("cuda:0")
for mb in (1, 4, 64):
    n = mb << 20
    pieces = max(1, 512 // mb)
    src = [torch.empty(n, dtype=torch.uint8).pin_memory() for _ in range(min(pieces, 8))]
    dst = [torch.empty(n, dtype=torch.uint8, device=dev) for _ in range(min(pieces, 8))]
    for streams in (1, 4):
        ss = [torch.cuda.Stream() for _ in range(streams)]
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for i in range(pieces):
                with torch.cuda.stream(ss[i % streams]):
                    dst[i % len(dst)].copy_(src[i % len(src)], non_blocking=True)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"{mb:3d} MB pieces x {pieces}, {streams} stream(s): {pieces * n / dt / 1e9:.1f} GB/s", flush=True)
src = torch.empty(256 << 20, dtype=torch.uint8)
d = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
torch.cuda.synchronize(); t0 = time.perf_counter(); d.copy_(src); torch.cuda.synchronize()
print(f"pageable 256 MB: {(256 << 20) / (time.perf_counter() - t0) / 1e9:.1f} GB/s")
