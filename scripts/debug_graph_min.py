"""Does `sync; eager kernel; graph.replay(); sync` stall on this ROCm for plain torch graphs?  (no geosplatting code)"""
import time, torch
dev = torch.device("cuda:0")
sync = torch.cuda.synchronize
x = torch.zeros(1 << 20, device=dev); y = torch.zeros(1 << 20, device=dev); one = torch.zeros(1024, device=dev)
pin = torch.zeros(2, dtype=torch.int64).pin_memory(); cnt = torch.zeros(2, dtype=torch.int64, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
import ctypes
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
z = torch.ones(1 << 16, device=dev)
def body(kind):
    x.add_(1)
    if kind == 4:                       # a real memset node (hipMemsetAsync), as gs_project_fwd / gs_isect_offsets issue them
        rc = hip.hipMemsetAsync(z.data_ptr(), 0, z.numel() * 4, torch.cuda.current_stream().cuda_stream); assert rc == 0, rc
        z.add_(1)
    if kind == 5:                       # memset on a forked stream
        cur = torch.cuda.current_stream(); s1.wait_stream(cur)
        with torch.cuda.stream(s1):
            rc = hip.hipMemsetAsync(z.data_ptr(), 0, z.numel() * 4, torch.cuda.current_stream().cuda_stream); assert rc == 0, rc
            z.add_(1)
            rc = hip.hipMemcpyAsync(pin.data_ptr(), cnt.data_ptr(), 16, 2, torch.cuda.current_stream().cuda_stream); assert rc == 0, rc
        cur.wait_stream(s1)
    if 1 <= kind <= 3:                  # fork / join over two side streams
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur); s2.wait_stream(cur)
        with torch.cuda.stream(s1):
            y.add_(2)
            if kind >= 2:
                pin.copy_(cnt, non_blocking=True)       # D2H into pinned memory as a graph node
        with torch.cuda.stream(s2):
            x.mul_(1.0)
            if kind >= 3:
                cnt.zero_()                              # memset node
        cur.wait_stream(s1); cur.wait_stream(s2)
    x.add_(y)
def run(name, fn):
    sync(); t0 = time.perf_counter(); fn(); sync(); print("  %-34s %9.2f ms" % (name, (time.perf_counter() - t0) * 1e3), flush=True)
for kind in range(6):
    w = torch.cuda.Stream(); w.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(w):
        body(kind)
    torch.cuda.current_stream().wait_stream(w); sync()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body(kind)
    print("graph kind", kind)
    run("replay", g.replay); run("replay", g.replay)
    run("eager kernel; replay", lambda: (one.add_(1), g.replay()))
    run("replay", g.replay)
    run("eager kernel; replay", lambda: (one.add_(1), g.replay()))
