"""Developer tool: occupancy timeline of k_mesh_raster_shade from per-workgroup wall-clock stamps (diagnostic build):
how many workgroups are in flight over time, and how much of the launch is the under-filled tail."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mvtn_amd import build as b
diag = os.path.join(ROOT, "mvtn_amd", "libmvr_hip_diag.so")
b.build_hip(out=diag, extra_flags=["-DMVR_DIAG", "-DMVR_DIAG_WALL"])
os.environ["MVR_LIB"] = diag
import numpy as np, torch
from mvtn_amd import MVRenderer, Meshes, PackedMeshes, _lib, synthetic as syn
lib = _lib.load()
dev = torch.device("cuda:0")
B, V, H = 16, 12, 224
vl, fl = zip(*[syn.synthetic_mesh("M", i) for i in range(B)])
packed = PackedMeshes.from_meshes(Meshes([torch.from_numpy(v) for v in vl], [torch.from_numpy(f) for f in fl]), dev)
az, el, di = syn.spherical_views(V)
ta, te, td = (torch.from_numpy(np.tile(x, (B, 1))).to(dev) for x in (az, el, di))
r = MVRenderer(V, H, pc_rendering=False, light_direction="fixed", rotation_check="off").eval()
for _ in range(3): r(packed, None, ta, te, td)
buf = torch.zeros((1 << 20) + 8 * 4096, dtype=torch.int64, device=dev)
lib.mvr_diag_set_buffer.argtypes = [ctypes.c_void_p]
lib.mvr_diag_set_buffer(buf.data_ptr())
r(packed, None, ta, te, td)
torch.cuda.synchronize()
nblocks = 8 * ((B * V + 7) // 8) * 28
d = buf.cpu().numpy().reshape(-1, 8)[:nblocks]
st, en = d[:, 0].astype(np.float64), np.maximum(d[:, 2], d[:, 5]).astype(np.float64)   # exact kernel: stamp 2; fast kernel: stamp 5
work = en > 0                      # non-empty strips have an end stamp
t0 = st[st > 0].min()
tick_us = 0.01                     # wall_clock64: 100 MHz
st, en = (st - t0) * tick_us, (en - t0) * tick_us
T = en[work].max()
print("kernel span (first start -> last end): %.1f us; %d working workgroups, %d empty" % (T, work.sum(), (~work).sum()))
busy = (en[work] - st[work]).sum()
print("sum of workgroup lifetimes / (1024 slots x span) = %.1f %%" % (100 * busy / (1024 * T)))
edges = np.linspace(0, T, 21)
for a, bb in zip(edges[:-1], edges[1:]):
    mid = 0.5 * (a + bb)
    active = ((st[work] <= mid) & (en[work] >= mid)).sum()
    print("  t = %5.1f us: %4d workgroups in flight" % (mid, active))
# per XCD (workgroup index & 7): when its last workgroup ends, and the sum of its workgroups' lifetimes
xcd = (np.arange(nblocks) & 7)[work]
for x in range(8):
    m = xcd == x
    print("  XCD %d: %4d working workgroups, lifetimes sum %.0f us (/128 slots = %.1f us), last end %.1f us, first start %.1f us" % (
        x, m.sum(), (en[work][m] - st[work][m]).sum(), (en[work][m] - st[work][m]).sum() / 128, en[work][m].max(), st[work][m].min()))
# what an ideal longest-first schedule of the measured workgroup lifetimes would take on each XCD's 128 slots
import heapq
for x in range(8):
    m = xcd == x
    dur = np.sort((en[work][m] - st[work][m]))[::-1]
    slots = [0.0] * 128
    heapq.heapify(slots)
    for d_ in dur:
        heapq.heappush(slots, heapq.heappop(slots) + d_)
    print("  XCD %d: longest-first makespan of the measured lifetimes %.1f us (sum / 128 = %.1f, measured last end %.1f)" % (
        x, max(slots), dur.sum() / 128, en[work][m].max()))
# list scheduling (next workgroup to the first free slot) of the measured lifetimes in three orders: as dispatched now
# (centre strips first), by the strip's listed faces (known after the pre-pass), by the lifetime itself (ideal)
gln = d[:, 6].astype(np.float64)
allidx = np.arange(nblocks)
def makespan(order):
    slots = [0.0] * 128
    heapq.heapify(slots)
    for i in order:
        heapq.heappush(slots, heapq.heappop(slots) + (en[i] - st[i] if work[i] else 0.05))
    return max(slots)
for x in range(8):
    idx = allidx[(allidx & 7) == x]
    life = np.where(work[idx], en[idx] - st[idx], 0.0)
    print("  XCD %d: simulated makespan: dispatch order %.1f us, by listed faces %.1f us, by lifetime %.1f us" % (
        x, makespan(idx), makespan(idx[np.argsort(-gln[idx], kind="stable")]), makespan(idx[np.argsort(-life, kind="stable")])))
