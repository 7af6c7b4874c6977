"""Host-link ceilings of the box for the PCIe-inclusive pipeline (bench.py: value_incl_pcie): pinned H2D / D2H of tile-sized buffers on
1, 2 and 4 copy streams, both directions at once, and H2D next to a busy GPU."""
import sys, time
import torch

dev = torch.device("cuda", 0)
n_tiles, tile_bytes = 32, 2 ** 28
host = [torch.empty(tile_bytes // 2, dtype=torch.int16, pin_memory=True) for _ in range(n_tiles)]
for h in host:
    h.fill_(3)
devt = [torch.empty(tile_bytes // 2, dtype=torch.int16, device=dev) for _ in range(n_tiles)]
torch.cuda.synchronize()


def run(direction, n_streams, busy=False, split=1):
    streams = [torch.cuda.Stream(device=dev) for _ in range(n_streams)]
    work = None
    if busy:
        a = torch.rand(8192, 8192, device=dev)
        work = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if busy:
        with torch.cuda.stream(work):
            for _ in range(40):
                a = a * 1.0001 + 0.5
    k = 0
    for i in range(n_tiles):
        step = (tile_bytes // 2) // split
        for s_ in range(split):
            sl = slice(s_ * step, (s_ + 1) * step)
            with torch.cuda.stream(streams[k % n_streams]):
                if direction in ("h2d", "both"):
                    devt[i][sl].copy_(host[i][sl], non_blocking=True)
                if direction == "d2h":
                    host[i][sl].copy_(devt[i][sl], non_blocking=True)
            if direction == "both":
                with torch.cuda.stream(streams[(k + 1) % n_streams]):
                    j = (i + n_tiles // 2) % n_tiles
                    host[j][sl].copy_(devt[j][sl], non_blocking=True)
            k += 1
    for s in streams:
        s.synchronize()
    dt = time.perf_counter() - t0
    torch.cuda.synchronize()
    gb = n_tiles * tile_bytes / 1e9 * (2 if direction == "both" else 1)
    print(f"{direction:5s} streams={n_streams} split={split} busy={int(busy)}: {dt * 1e3:7.1f} ms  {gb / dt:6.1f} GB/s", flush=True)


for rep in range(2):
    run("h2d", 1); run("h2d", 2); run("h2d", 4); run("h2d", 2, split=2); run("h2d", 1, busy=True); run("h2d", 2, busy=True)
    run("d2h", 1); run("d2h", 2); run("both", 2); run("both", 4)
