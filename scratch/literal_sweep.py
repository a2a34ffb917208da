"""configs[1] literal launch (256 rays x 10 000 triangles): kernel time vs the number of blocks the launcher
aims for (DRT_DENSE_BLOCKS), measured inside a HIP graph of 200 launches.  One process per setting."""
import os, subprocess, sys
if len(sys.argv) > 1:
    import numpy as np, torch
    sys.path.insert(0, ".")
    import differt_amd._lib as lib
    from differt_amd._tensors import ptr, stream
    from bench import make_cfg2
    R, T = int(sys.argv[1]), 10000
    o, d, tv = (torch.as_tensor(x, device="cuda") for x in make_cfg2(R, T, seed=99))
    t = torch.empty((R, T), device="cuda"); h = torch.empty((R, T), dtype=torch.uint8, device="cuda")
    eps = 10 * 1.1920929e-7
    def step(): lib.call("drt_ray_intersect_triangle_dense", ptr(o), ptr(d), R, ptr(tv), T, eps, ptr(t), ptr(h), stream())
    for _ in range(20): step()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(200): step()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): g.replay()
    e1.record(); torch.cuda.synchronize()
    print(os.environ.get("DRT_DENSE_BLOCKS", "640"), R, "us/launch", e0.elapsed_time(e1) * 1e3 / 2000)
else:
    for R in (256, 1024, 4096):
        for b in (320, 640, 1280, 1792, 2560, 5120):
            subprocess.run([sys.executable, __file__, str(R)], env=dict(os.environ, DRT_DENSE_BLOCKS=str(b)))
