"""Developer tool (uses the oracle as the checker, hence under tests/): prints where GPU and oracle differ."""
import sys
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(REPO)); sys.path.insert(0, str(REPO / "sycl-ray-tracer_amd"))
import numpy as np
from oracle import oracle as O
from rtamd import abi, scenes
from rtamd.renderer import Scene
name = sys.argv[1] if len(sys.argv) > 1 else "cube"
sd = scenes.get_scene(name)
gs, osc = Scene(sd, 0), O.OracleScene(sd)
rng = np.random.default_rng(5)
n = 20000
tw = sd.world_triangles()
lo, hi = tw.reshape(-1, 3).min(0), tw.reshape(-1, 3).max(0)
org = rng.uniform(lo - 0.1 * (hi - lo) - 0.5, hi + 0.1 * (hi - lo) + 0.5, (n, 3)).astype(np.float32)
dirs = rng.normal(size=(n, 3)).astype(np.float32)
dirs[: n // 4] *= 1e-2
dirs = dirs.astype(np.float16).astype(np.float32)
k = n // 3
ti = rng.integers(0, sd.n_triangles, k)
b = rng.dirichlet((1, 1, 1), k)
org[:k] = np.einsum("ij,ijk->ik", b, tw[ti]).astype(np.float32)
dirs[k: k + 60] = np.tile(np.array([[1, 0, 0], [0, -1, 0], [0, 0, 1], [0, 1, 1], [-1, 0, 1], [1, 1, 0]], np.float32), (10, 1))
g = gs.intersect(org, dirs); e = osc.intersect(org, dirs, use_bvh=False)
bad = np.nonzero((g[3] != e[3]) | (g[0] != e[0]) | (g[1] != e[1]) | (g[2] != e[2]))[0]
print("mismatches", bad.size, "of", n, "k=", k)
np.set_printoptions(precision=9, floatmode="unique")
for i in bad[:20]:
    print(i, "org", org[i].view(np.uint32), org[i], "dir", dirs[i].view(np.uint32), dirs[i])
    print("   gpu", g[0][i], g[1][i], g[2][i], g[3][i], " cpu", e[0][i], e[1][i], e[2][i], e[3][i])
    if i < k: print("   started on tri", ti[i])
