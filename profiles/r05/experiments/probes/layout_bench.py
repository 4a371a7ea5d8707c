"""Extraction (and sampler) speed on one 512^3 volume in the two memory layouts: x fastest, and the C# float[,,] order (z fastest)."""
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, numpy as np
import volumetricterrain_amd as vt
n = 512
d = n + 2
ex = vt.Extractor(0)
buf = torch.empty(d ** 3, dtype=torch.float32, device="cuda")
prm = vt.density_params("perlin3d", 1024)
org = np.zeros((1, 3), np.int32)
for name, strides in (("x fastest", (1, d, d * d)), ("z fastest (C# float[,,])", (d * d, d, 1))):
    for _ in range(3):
        ex.density_fill_device(prm, org, (d, d, d), strides, d ** 3, buf.data_ptr())
    for _ in range(2):
        T = ex.extract_volumes_device(buf.data_ptr(), (n, n, n), strides, 1, d ** 3)
    ms = []
    for _ in range(5):
        T = ex.extract_volumes_device(buf.data_ptr(), (n, n, n), strides, 1, d ** 3)
        ms.append(ex.last_stage_ms())
    m = {k: sorted(x[k] for x in ms)[2] for k in ms[0]}
    print("%-28s T=%d  fill %.3f ms  classify %.3f  scan %.3f  emit %.3f  total %.3f ms" % (name, T, ex.last_fill_ms(), m["classify"], m["scan"], m["emit"], m["total"]))
