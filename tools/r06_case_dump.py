"""Round 6 debugging aid: run the stress case 61760 through the C ABI with a workspace of its own and read the fast rasteriser's settle
candidates of one pixel back from the gap of its strip's list region.    python3 tools/r06_case_dump.py [view] [row] [col]"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from mvtn_amd import functional as F, _lib
from mvtn_amd.structures import Meshes, PackedMeshes
view, row, col = (int(x) for x in (sys.argv[1:4] + ["1", "4", "8"])[:3])
d = np.load(os.path.join(ROOT, "tests", "golden", "stress_case_61760.npz"))
vl, fl = [d["v0"], d["v1"]], [d["f0"], d["f1"]]
az, el, di, S = d["az"], d["el"], d["di"], int(d["size"])
dev = torch.device("cuda:0")
lib = _lib.load()
t = lambda x: torch.as_tensor(np.ascontiguousarray(x), dtype=torch.float32, device=dev)
packed = PackedMeshes.from_meshes(Meshes(vl, fl), dev)
R, T = F.look_at_view_transform(t(di.reshape(-1)), t(el.reshape(-1)), t(az.reshape(-1)))
params = F.mesh_params(2, S, 1, (0.9, 0.8, 0.7), (0.1, 0.2, 0.3), z_clip=0.0)
mb = packed.c_struct()
nbytes = lib.mvr_mesh_workspace_bytes(mb, params)
ws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
B, V, N, TH = 2, 2, 4, 8
img = torch.empty(B, V, 3, S, S, device=dev); p2f = torch.empty(N, S, S, dtype=torch.int32, device=dev)
light = t([[0, 1, 0]])
st = torch.cuda.current_stream().cuda_stream
_lib.check(lib.mvr_render_meshes_forward(mb, params, R.data_ptr(), T.data_ptr(), light.data_ptr(), img.data_ptr(), p2f.data_ptr(), ws.data_ptr(), nbytes, st), "fwd")
torch.cuda.synchronize()
al = lambda x: (x + 255) // 256 * 256
TV, mf = packed.total_verts, packed.max_faces
n_strips = (S + TH - 1) // TH
a64 = lambda x: (x + 63) // 64 * 64
cap = a64(mf // 2) + a64(TH * S // 4); cap = max(cap, a64(min(mf, 256) + TH * S)); cap = min(max(cap, 256), 3584)
off = al(3 * TV * 8); off += al(2 * TV * 16); off += al(V * TV * 16)
stride = (mf + 7) // 8 * 8
off += al(N * stride * 2); off += al(N * n_strips * 18 * 4)
o_slist = off; off += al(N * n_strips * cap * 16)
o_front = off; off += al(N * n_strips * 4)
off += al(N * 8 * 4)
o_redo = off
w = ws.cpu().numpy()
strip = row // TH
sidx = view * n_strips + strip
front = int(w[o_front + 4 * sidx: o_front + 4 * sidx + 4].view(np.int32)[0])
redo = int(w[o_redo + 4 * sidx: o_redo + 4 * sidx + 4].view(np.uint32)[0])
near, far = front & 0x1fff, (front >> 13) & 0x1fff
print("workspace %d B, cap %d; view %d strip %d: near %d far %d flags %x; redo word: redone %d, candidates %d, whole-wave %d" % (nbytes, cap, view, strip, near, far, front >> 26, redo & 1, (redo >> 1) & 0x7fff, redo >> 16))
reg = w[o_slist + sidx * cap * 16: o_slist + (sidx + 1) * cap * 16]
ents = reg.view(np.int32).reshape(cap, 4)
lst = np.concatenate([ents[:near], ents[cap - far:]])     # list index e -> entry
gap = reg[near * 16:(cap - far) * 16].view(np.uint32).reshape(-1, 2)
n_amb = (redo >> 1) & 0x7fff
cand = gap[:n_amb]
pix = (row - strip * TH) * S + col
sel = cand[(cand[:, 0] & 0x1fff) == pix]
print("pixel %d: %d of %d candidates; library face %d" % (pix, len(sel), n_amb, int(p2f[view, row, col])))
for c in sel:
    e = int(c[0] >> 13)
    print("   list index %d -> face %d, bound %g" % (e, int(lst[e][0]) & 0x3fffffff if e < len(lst) else -1, float(np.array([c[1]], np.uint32).view(np.float32)[0])))
faces_here = sorted(set(int(lst[int(c[0] >> 13)][0]) & 0x3fffffff for c in cand if int(c[0] >> 13) < len(lst)))
print("faces among this strip's candidates:", faces_here[:40])
print("list (face ids in list order):", [int(x[0]) & 0x3fffffff for x in lst][:100])
# debug records of a -DEBUG variant (tools/lib_varC.so): 4 entries per survivor i at the end of the gap
allg = reg[near * 16:(cap - far) * 16].view(np.uint32).reshape(-1, 2)
for j in range(len(allg) - 4, 0, -4):
    r = allg[j:j + 4]
    if (int(r[0][0]) >> 28) == 0xD:
        i = int(r[0][0]) & 0xfffffff
        e, ef = int(r[0][1]) >> 16, int(r[0][1]) & 0xffff
        f = lambda ee: int(lst[ee][0]) & 0x3fffffff
        pz = lambda hi: float(np.array([(~hi) & 0xffffffff], np.uint32).view(np.float32)[0])
        print("survivor %d: e %d (face %d) key %08x %08x (pz %g)  old %08x %08x  ef %d (face %d) kf %08x %08x (pz %g)" % (
            i, e, f(e), int(r[1][0]), int(r[1][1]), pz(int(r[1][0])), int(r[2][0]), int(r[2][1]), ef, f(ef), int(r[3][0]), int(r[3][1]), pz(int(r[3][0]))))
