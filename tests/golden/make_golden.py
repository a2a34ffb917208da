#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the reference's own test DATA.

Run in the build container (needs /root/reference, which never travels to the GPU box):

    python tests/golden/make_golden.py

Outputs (committed):
  two_buildings.json      vertices / triangles of differt/tests/geometry/two_buildings.obj read the
                          way differt-core/src/geometry/mesh.rs:399-429 reads it (all vertices kept,
                          every non-triangle face skipped).
  reference_goldens.json  the known-answer tables of the reference's tests, as plain numbers, each
                          with the file:line it comes from.

  bruxelles.npz, manhattan.npz, manhattan_small.npz
                          geometry arrays (vertices f32[V,3], triangles i32[T,3]) of the reference's in-tree
                          meshes docs/source/notebooks/*.obj read by the same rule -- bruxelles.obj is the mesh
                          of the reference's own benchmark harness (differt/tests/benchmarks/fixtures.py:43-68).

Fixtures are data (inputs + expected outputs), never reference source text.
"""

from __future__ import annotations

import json
from pathlib import Path

import numpy as np

REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent


def read_obj_triangles_only(path: Path):
    """OBJ subset reader mirroring mesh.rs:399-429: keep all `v`, keep `f` only if 3 indices."""
    vertices, triangles, skipped = [], [], 0
    for line in path.read_text().splitlines():
        parts = line.split()
        if not parts:
            continue
        if parts[0] == "v":
            vertices.append([float(x) for x in parts[1:4]])
        elif parts[0] == "f":
            idx = [int(tok.split("/")[0]) - 1 for tok in parts[1:]]
            if len(idx) == 3:
                triangles.append(idx)
            else:
                skipped += 1
    return vertices, triangles, skipped


def main() -> None:
    v, t, skipped = read_obj_triangles_only(REF / "differt/tests/geometry/two_buildings.obj")
    (OUT / "two_buildings.json").write_text(
        json.dumps(
            {
                "source": "differt/tests/geometry/two_buildings.obj via the rule of "
                "differt-core/src/geometry/mesh.rs:399-429",
                "skipped_non_triangle_faces": skipped,
                "vertices": v,
                "triangles": t,
            }
        )
    )

    for name in ("bruxelles", "manhattan", "manhattan_small"):
        mv, mt, msk = read_obj_triangles_only(REF / "docs/source/notebooks" / f"{name}.obj")
        # float32 as Mesh.from_core makes them (ME: jnp.asarray of the core mesh's f32 vertices), int32 indices
        np.savez_compressed(OUT / f"{name}.npz", vertices=np.asarray(mv, np.float32), triangles=np.asarray(mt, np.int32),
                            skipped_non_triangle_faces=np.int32(msk))
        print(f"wrote {OUT / (name + '.npz')} ({len(mv)} vertices, {len(mt)} triangles, {msk} skipped)")

    goldens = {
        # differt/tests/geometry/fixtures.py:64-71
        "advanced_path_tracing_example": {
            "source": "differt/tests/geometry/fixtures.py:64-71, differt/tests/geometry/test_scene.py:116-160",
            "tx": [0.0, 4.9352, 22.0],
            "rx": [0.0, 10.034, 1.50],
            "rtol": 1e-6,
            "orders": {
                "0": {"path_vertices": [[]], "objects": [[0, 0]]},
                "1": {
                    "path_vertices": [[[-0.06917738914489746, 14.946798324584961, 8.24851131439209]]],
                    "objects": [[0, 8, 0]],
                },
                "2": {
                    "path_vertices": [
                        [
                            [-0.125960111618042, 14.946202278137207, 13.787875175476074],
                            [-0.04232808202505112, 5.0, 5.629261016845703],
                        ]
                    ],
                    "objects": [[0, 9, 22, 0]],
                },
                "3": {
                    "path_vertices": [
                        [
                            [-0.17936798930168152, 14.945640563964844, 16.1051082611084],
                            [-0.14879928529262543, 5.0, 10.249288558959961],
                            [-0.11822860687971115, 14.946282386779785, 4.393090724945068],
                        ]
                    ],
                    "objects": [[0, 9, 22, 8, 0]],
                },
                "4": {
                    "path_vertices": [
                        [
                            [-0.233406662940979, 14.945074081420898, 17.426870346069336],
                            [-0.25651583075523376, 5.0, 12.884565353393555],
                            [-0.2796238660812378, 14.944588661193848, 8.342482566833496],
                            [-0.09397590905427933, 5.0, 3.799619674682617],
                        ]
                    ],
                    "objects": [[0, 9, 23, 8, 22, 0]],
                },
            },
        },
        # differt/tests/geometry/test_utils.py:448-476
        "generate_all_path_candidates": {
            "source": "differt/tests/geometry/test_utils.py:448-476 (compared after a lexsort) and "
            "differt-core/src/geometry/graph.rs:1488-1513 (ordered)",
            "cases": [
                {"num_primitives": 0, "order": 0, "shape": [1, 0], "rows": [[]]},
                {"num_primitives": 8, "order": 0, "shape": [1, 0], "rows": [[]]},
                {"num_primitives": 0, "order": 5, "shape": [0, 5], "rows": []},
                {"num_primitives": 3, "order": 1, "shape": [3, 1], "rows": [[0], [1], [2]]},
                {
                    "num_primitives": 3,
                    "order": 2,
                    "shape": [6, 2],
                    "rows": [[0, 1], [0, 2], [1, 0], [1, 2], [2, 0], [2, 1]],
                },
                {
                    "num_primitives": 3,
                    "order": 3,
                    "shape": [12, 3],
                    "rows": [
                        [0, 1, 0], [0, 1, 2], [0, 2, 0], [0, 2, 1], [1, 0, 1], [1, 0, 2],
                        [1, 2, 0], [1, 2, 1], [2, 0, 1], [2, 0, 2], [2, 1, 0], [2, 1, 2],
                    ],
                },
            ],
        },
        # differt/tests/geometry/test_utils.py:555-577
        "ray_intersect_triangle_hits": {
            "source": "differt/tests/geometry/test_utils.py:555-577",
            "triangle": [[0.0, 0.0, 0.0], [1.0, 0.0, 0.0], [0.0, 1.0, 0.0]],
            "cases": [
                {"orig": [0.5, 0.5, 1.0], "dest": [0.5, 0.5, -1.0], "expected": True},
                {"orig": [0.0, 0.0, 1.0], "dest": [1.0, 1.0, -1.0], "expected": True},
                {"orig": [0.5, 0.5, 1.0], "dest": [0.5, 0.5, 0.5], "expected": False},
                {"orig": [0.5, 0.5, 1.0], "dest": [1.0, 1.0, 1.0], "expected": False},
                {"orig": [0.5, 0.5, 1.0], "dest": [1.0, 1.0, 1.5], "expected": False},
            ],
        },
        # differt/tests/geometry/test_utils.py:580-606
        "ray_intersect_triangle_t_and_hit": {
            "source": "differt/tests/geometry/test_utils.py:580-606",
            "ray_origin": [0.5, 0.5, -1.0],
            "ray_directions": [[0.0, 0.0, 1.0], [0.0, 0.0, 0.5], [0.0, 0.0, -1.0], [1.0, 0.0, 0.0]],
            "triangle_vertices": [
                [[0.0, 0.0, 0.0], [1.0, 0.0, 0.0], [0.0, 1.0, 0.0]],
                [[0.0, 0.0, 1.0], [1.0, 0.0, 1.0], [0.0, 1.0, 1.0]],
            ],
            "expected_t": [[1.0, 2.0], [2.0, 4.0], [-1.0, -2.0], [0.0, 0.0]],
            "expected_hit": [[True, True], [True, True], [False, False], [False, False]],
        },
        # differt/tests/geometry/test_image_method.py:19-29
        "image_of_vertex": {
            "source": "differt/tests/geometry/test_image_method.py:19-29",
            "vertices": [[0.0, 0.0, 1.0], [1.0, 2.0, 3.0]],
            "mirror_vertices": [[0.0, 0.0, 0.0]],
            "mirror_normals": [[0.0, 0.0, 1.0]],
            "expected": [[0.0, 0.0, -1.0], [1.0, 2.0, -3.0]],
        },
        # differt/tests/geometry/test_image_method.py:70-129
        "intersection_of_ray_with_plane": {
            "source": "differt/tests/geometry/test_image_method.py:70-129",
            "ray_origins": [[-1.0, 1.0, 0.0], [-2.0, 1.0, 0.0], [-3.0, 1.0, 0.0]],
            "ray_end": [2.0, -1.0, 0.0],
            "cases": [
                {
                    "plane_vertex": [0.0, 0.0, 0.0],
                    "plane_normal": [0.0, 1.0, 0.0],
                    "expected": [[0.5, 0.0, 0.0], [0.0, 0.0, 0.0], [-0.5, 0.0, 0.0]],
                },
                {"plane_vertex": [0.0, 0.0, -1.0], "plane_normal": [0.0, 0.0, 1.0], "expected": "inf"},
                {"plane_vertex": [0.0, 0.0, 0.0], "plane_normal": [0.0, 0.0, 1.0], "expected": "origins"},
            ],
        },
        # differt/tests/geometry/fixtures.py:82-117
        "planar_mirrors_setup": {
            "source": "differt/tests/geometry/fixtures.py:82-117 (used by test_image_method.py:160-219)",
            "from_vertex": [0.0, 0.0, 0.0],
            "to_vertex": [1.0, 0.0, 0.0],
            "mirror_vertices": [[0.0, 1.0, 0.0], [0.0, -1.0, 0.0], [0.0, 1.0, 0.0], [0.0, -1.0, 0.0]],
            "mirror_normals": [[0.0, -1.0, 0.0], [0.0, 1.0, 0.0], [0.0, -1.0, 0.0], [0.0, 1.0, 0.0]],
            "paths": [[0.125, 1.0, 0.0], [0.375, -1.0, 0.0], [0.625, 1.0, 0.0], [0.875, -1.0, 0.0]],
        },
        # differt/src/differt/geometry/_mesh.py:2172-2217 + tests/geometry/test_utils.py:440-445
        "box_with_top": {
            "source": "differt/src/differt/geometry/_mesh.py:2172-2217 (Mesh.box(with_top=True): 12 triangles)",
            "triangles": [
                [0, 1, 2], [0, 2, 3], [3, 2, 4], [3, 4, 5], [5, 4, 6], [5, 6, 7], [7, 6, 1], [7, 1, 0],
                [1, 4, 2], [1, 6, 4], [0, 3, 5], [0, 5, 7],
            ],
        },
        # differt-core/tests/geometry/test_graph.py:158-205 ; graph.rs:356-362
        "complete_graph_counts": {
            "source": "differt-core/src/geometry/graph.rs:314-377; differt-core/tests/geometry/test_graph.py:158-205",
            "formula": "from,to not in graph: n*(n-1)**(depth-3) for depth>=3; depth==2 -> 1",
        },
        # differt/tests/geometry/test_mesh.py:2004-2073 (first-hit on Mesh.box(2,2,2))
        "first_hit_box": {
            "source": "differt/tests/geometry/test_mesh.py:2004-2073",
            "box": [2.0, 2.0, 2.0],
            "ray_origins": [[0.0, 0.0, 3.0], [0.0, 3.0, 0.0], [3.0, 0.0, 0.0]],
            "ray_directions": [[0.0, 0.0, -1.0], [0.0, -1.0, 0.0], [-1.0, 0.0, 0.0]],
            "note": "the reference only asserts Warp == pure-JAX (indices, t, Jacobians at 1e-5); no literal values",
        },
    }
    (OUT / "reference_goldens.json").write_text(json.dumps(goldens, indent=1))
    print(f"wrote {OUT / 'two_buildings.json'} ({len(v)} vertices, {len(t)} triangles, {skipped} skipped)")
    print(f"wrote {OUT / 'reference_goldens.json'}")


if __name__ == "__main__":
    main()
