"""pageable vs pinned host<->device copy rates (what vm_align_batch's uploads / downloads of host blobs pay)"""
import time, torch
n = 150 << 20
d = torch.empty(n, dtype=torch.uint8, device='cuda')
for name, h in (('pageable', torch.empty(n, dtype=torch.uint8)), ('pinned', torch.empty(n, dtype=torch.uint8).pin_memory())):
    h.fill_(1)
    for direction in ('h2d', 'd2h'):
        torch.cuda.synchronize(); ts = []
        for _ in range(5):
            t = time.time()
            (d.copy_(h, non_blocking=True) if direction == 'h2d' else h.copy_(d, non_blocking=True)); torch.cuda.synchronize()
            ts.append(time.time() - t)
        print('%s %s: %.1f ms for 150 MB = %.1f GB/s' % (name, direction, min(ts) * 1e3, n / min(ts) / 1e9))
