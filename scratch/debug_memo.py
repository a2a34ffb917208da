import sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import torch, differt_amd.geometry as G
box = G.Mesh.box(2.0, 2.0, 2.0, with_top=True)
tx = torch.tensor([[5.0, 0.3, 0.1]], device="cuda"); rx = torch.tensor([[-5.0, 0.2, 0.3]], device="cuda")
s = G.HybridPathTracer(num_rays=50_000)
sc = G.Scene(tx, rx, box)
print(box._handle_key())
a = s._visible_sets(sc)
print(box._handle_key()); print(s._vis_cache[0][:4])
b = s._visible_sets(sc)
print(a is b, sc.mesh is box, type(sc.transmitters))
print(box._handle_key()); print(s._vis_cache[0][:4])
