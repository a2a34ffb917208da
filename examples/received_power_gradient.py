"""End-to-end differentiable radio propagation on an MI355X: move a transmitter uphill on the received power.

    python examples/received_power_gradient.py [steps]

Scene: a synthetic 100-building Manhattan mesh (1000 triangles, concrete), one transmitter above a roof,
sixteen receivers at street level.  Every step
  1. traces the order-0/1/2 paths with the per-pair visibility-pruned tracer (HybridPathTracer.trace_pairs:
     candidates unranked on the GPU, no table),
  2. turns every valid path into a complex channel coefficient (differt_amd.plugins.deepmimo.paths_channel:
     s/p bases, Fresnel coefficients, spreading, phase),
  3. sums the received power over paths and receivers (incoherently) and back-propagates through the channel
     kernel (forward-mode duals) and the image-method tracer (hand-written VJP) down to the transmitter
     position, then takes a gradient-ascent step.
Everything between the NumPy inputs and the printed numbers runs in HIP kernels behind the C ABI.
"""

from __future__ import annotations

import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import differt_amd.geometry as G  # noqa: E402
import synthetic_scenes as S  # noqa: E402
from differt_amd.plugins import deepmimo  # noqa: E402


def main(steps: int = 5, frequency: float = 3.5e9) -> list[float]:
    V, Tr, centres, heights = S.manhattan(100, seed=7)
    tx0, rx = S.manhattan_tx_rx(centres, heights, 1, 16, seed=3)
    mesh = G.Mesh(V, Tr).set_materials("itu_concrete")
    n_tab, thickness = deepmimo.material_tables(mesh.material_names, deepmimo.materials, frequency)
    solver = G.HybridPathTracer(num_rays=100_000, accel="bvh", sample_triangles=True)
    tx = torch.tensor(tx0, device="cuda", requires_grad=True)
    rxs = torch.tensor(rx, device="cuda")
    history = []
    for step in range(steps):
        scene = G.Scene(tx, rxs, mesh)
        total = torch.zeros((), device="cuda")
        npaths = 0
        for order in (0, 1, 2):
            paths = solver.trace_pairs(scene, order)
            if paths.objects.shape[0] == 0:
                continue
            ch = deepmimo.paths_channel(paths, mesh, n_tab, thickness, frequency)
            total = total + (ch["a"].abs() ** 2).sum()  # incoherent sum of |a|^2 over paths and receivers
            npaths += int(paths.objects.shape[0])
        power_db = 10.0 * torch.log10(total / 376.73031341259)
        (grad,) = torch.autograd.grad(power_db, tx)
        history.append(float(power_db.detach()))
        print(f"step {step}: {npaths} paths, total received power {history[-1]:8.3f} dBW, "
              f"|d power / d tx| = {float(grad.norm()):.4f} dB/m, tx = {tx.detach().cpu().numpy().round(3).tolist()}")
        with torch.no_grad():
            tx += 0.5 * grad / (grad.norm() + 1e-12)  # half a metre uphill
    return history


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 5)
