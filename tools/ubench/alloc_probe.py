import time, torch, ctypes
hip = ctypes.CDLL('libamdhip64.so')
def hipmalloc(n):
    p = ctypes.c_void_p(); t=time.time(); rc = hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(n)); dt=time.time()-t; return p, rc, dt
torch.cuda.init(); torch.zeros(1, device='cuda'); torch.cuda.synchronize()
for gb in (1, 8, 32, 32, 1, 8):
    p, rc, dt = hipmalloc(gb<<30)
    t=time.time(); hip.hipMemset(p, 0, ctypes.c_size_t(gb<<30)); hip.hipDeviceSynchronize(); dm=time.time()-t
    t=time.time(); hip.hipFree(p); df=time.time()-t
    print('hipMalloc %2d GB: %.3f s (rc %d)  memset %.3f s  free %.3f s' % (gb, dt, rc, dm, df))
# many small allocations
t=time.time(); ps=[hipmalloc(64<<20)[0] for _ in range(200)]; print('200 x 64 MB: %.3f s' % (time.time()-t))
