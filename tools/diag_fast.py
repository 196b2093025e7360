"""Developer tool: phase times of k_mesh_raster_fast (-DMVR_DIAG build: per-workgroup s_memtime stamps of wave 0 into a
buffer of their own), C2 workload.     python3 tools/diag_fast.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
diag = os.path.join(ROOT, "mvtn_amd", "libmvr_hip_diag.so")
from mvtn_amd import build as b  # noqa: E402
b.build_hip(out=diag, extra_flags=["-DMVR_DIAG"] + [x for x in os.environ.get("MVR_DIAG_EXTRA", "").split() if x])
os.environ["MVR_LIB"] = diag
import numpy as np  # noqa: E402
import torch  # noqa: E402
from mvtn_amd import MVRenderer, Meshes, PackedMeshes, _lib, synthetic as syn  # noqa: E402
lib = _lib.load()
dev = torch.device("cuda:0")
B, V, H = 16, 12, 224
TIER = os.environ.get("MVR_TIER", "M")     # MVR_TIER=L: the C5 shard (8 tier-L meshes x 20 views)
if TIER == "L":
    B, V = 8, 20
GEO = os.environ.get("MVR_GEOMETRY", "ico")   # ico / cad / ragged (bench.py --geometry)
SEED0 = int(os.environ.get("MVR_SEED0", "0"))
vl, fl = zip(*[syn.geometry_mesh(GEO, TIER, SEED0 + i) for i in range(B)])
packed = PackedMeshes.from_meshes(Meshes([torch.from_numpy(v) for v in vl], [torch.from_numpy(f) for f in fl]), dev)
az, el, di = syn.spherical_views(V)
ta, te, td = (torch.from_numpy(np.tile(x, (B, 1))).to(dev) for x in (az, el, di))
r = MVRenderer(V, H, pc_rendering=False, light_direction="fixed", rotation_check="off").eval()
for _ in range(3):
    r(packed, None, ta, te, td)
nblocks = 8 * ((B * V + 7) // 8) * 28
buf = torch.zeros((1 << 20) + 8 * 4096, dtype=torch.int64, device=dev)
lib.mvr_diag_set_buffer.argtypes = [ctypes.c_void_p]
lib.mvr_diag_set_buffer(buf.data_ptr())
r(packed, None, ta, te, td)
torch.cuda.synchronize()
raw = buf.cpu().numpy().reshape(-1, 8)[:nblocks]
raw = raw[raw[:, 5] > 0]
d = raw.astype(np.float64)
d[:, 6] = raw[:, 6] & 0xffffffff
d[:, 7] = raw[:, 7] & 0xffffffff
n_wide, n_fat = (raw[:, 6] >> 32).astype(np.float64), (raw[:, 7] >> 32).astype(np.float64)
tot = d[:, 5] - d[:, 0]
print("workgroups with work: %d   cycles/workgroup mean %.0f max %.0f; listed faces mean %.0f; ambiguous pixels total %d" % (
    len(d), tot.mean(), tot.max(), d[:, 6].mean(), d[:, 7].sum()))
print("cycles/workgroup percentiles 50 / 90 / 99 / max: %.0f / %.0f / %.0f / %.0f; sum of workgroup cycles %.3g (a CU holds 4: / 1024 / 2.1e9 = %.1f us if perfectly packed)" % (
    np.percentile(tot, 50), np.percentile(tot, 90), np.percentile(tot, 99), tot.max(), tot.sum(), tot.sum() / 1024 / 2.1e9 * 1e6))
# per XCD (blockIdx & 7: an XCD owns a contiguous range of views) and per object: is the launch balanced across the chip?
allr = buf.cpu().numpy().reshape(-1, 8)[:nblocks]
tot_all = np.where(allr[:, 5] > 0, allr[:, 5] - allr[:, 0], 0).astype(np.float64)
xs = np.array([tot_all[x::8].sum() for x in range(8)])
print("summed workgroup cycles per XCD (M): %s   max / mean %.3f" % (" ".join("%.1f" % (x / 1e6) for x in xs), xs.max() / xs.mean()))
# per object (the launch maps bid -> xcd = bid & 7, j = bid >> 3, view = xcd * vpx + j % vpx)
vpx = (B * V + 7) // 8
bid = np.arange(nblocks)
view_of = (bid & 7) * vpx + (bid >> 3) % vpx
obj = np.array([tot_all[(view_of // V) == b].sum() for b in range(B)])
print("summed workgroup cycles per object (M): %s" % " ".join("%.1f" % (x / 1e6) for x in obj))
srt = np.sort(obj)[::-1]
bins8 = np.zeros(8)
for x in srt:
    bins8[np.argmin(bins8)] += x
print("  objects dealt to 8 XCDs heaviest-first onto the lightest XCD: max / mean %.3f (as launched: %.3f)" % (bins8.max() / bins8.mean(), xs.max() / xs.mean()))
# occupancy over time per XCD (stamps of one XCD share a clock): how long does the launch run with fewer than half / a quarter of
# its workgroup slots busy — the tail —, and when does the busiest period end?
for x in range(8):
    r = allr[x::8]
    r = r[r[:, 5] > 0]
    if not len(r):
        continue
    t0x = r[:, 0].min()
    ev = np.concatenate([np.stack([r[:, 0] - t0x, np.ones(len(r))], 1), np.stack([r[:, 5] - t0x, -np.ones(len(r))], 1)])
    ev = ev[np.argsort(ev[:, 0], kind="stable")]
    occ = np.cumsum(ev[:, 1])
    T = ev[-1, 0]
    peak = occ.max()
    below_half = ev[np.where(occ < 0.5 * peak)[0], 0]
    below_half = below_half[below_half > 0.3 * T]
    below_q = ev[np.where(occ < 0.25 * peak)[0], 0]
    below_q = below_q[below_q > 0.3 * T]
    dt = np.diff(ev[:, 0])
    avg_occ = (occ[:-1] * dt).sum() / T
    u = 0.01 if "MVR_DIAG_WALL" in os.environ.get("MVR_DIAG_EXTRA", "") else 1.0 / 2100.0   # microseconds per stamp unit (wall clock: 100 MHz; else ~2.1 GHz, NOT comparable across CUs)
    print("XCD %d: span %.1f us, peak %d workgroups resident, mean %.1f (%.0f %% of the peak); occupancy below half of the peak from %.1f us (the last %.0f %% of the span), below a quarter from %.1f us" % (
        x, T * u, peak, avg_occ, 100 * avg_occ / peak, (below_half.min() if len(below_half) else T) * u, 100 * (1 - (below_half.min() if len(below_half) else T) / T), (below_q.min() if len(below_q) else T) * u))
if os.environ.get("MVR_DIAG_DUMP"):   # start / end stamps of every workgroup (blockIdx order) for offline scheduling simulations
    np.save(os.environ["MVR_DIAG_DUMP"], allr)
t_first, t_last = allr[allr[:, 5] > 0, 0].min(), allr[:, 5].max()
ends = np.array([allr[x::8, 5].max() - t_first for x in range(8)])
print("kernel span by the stamps: %.0f cycles; last workgroup of each XCD ends at (k cycles): %s" % (t_last - t_first, " ".join("%.0f" % (e / 1e3) for e in ends)))
two = d[:, 2] > 0
names = ["init (tables, key clear, first entries)", "raster (all segments, wave 0)", "wait for the other waves", "exact settle of the candidates + resolve + shade"]
spans = [d[:, 1] - d[:, 0], d[:, 3] - d[:, 1], d[:, 4] - d[:, 3], d[:, 5] - d[:, 4]]
for nme, sp in zip(names, spans):
    print("  %-42s %5.1f %%   mean %7.0f cycles" % (nme, 100 * sp.sum() / tot.sum(), sp.mean()))
print("  two-segment workgroups: %.1f %%; their near segment (wave 0): mean %.0f cycles of raster %.0f" % (
    100 * two.mean(), (d[two, 2] - d[two, 1]).mean(), (d[two, 3] - d[two, 1]).mean()))
nl = d[:, 6]
for lo, hi in [(1, 65), (65, 129), (129, 257), (257, 385), (385, 513), (513, 100000)]:
    m = (nl >= lo) & (nl < hi)
    if m.any():
        print("    faces [%4d,%5d): %5.1f %% of wgs  total %7.0f  init %6.0f  raster %7.0f  wait %6.0f  resolve %7.0f" % (
            lo, hi, 100 * m.mean(), tot[m].mean(), spans[0][m].mean(), spans[1][m].mean(), spans[2][m].mean(), spans[3][m].mean()))

rs = buf.cpu().numpy().reshape(-1, 8)[(1 << 19) // 8:(1 << 19) // 8 + nblocks].astype(np.float64)
rs = rs[rs[:, 5] > 0]
if len(rs):
    print("resolve+shade phases (wave 0 stamps), workgroups %d: covered px mean %.0f, winner faces mean %.0f max %.0f" % (len(rs), rs[:, 6].mean(), rs[:, 7].mean(), rs[:, 7].max()))
    for nme, a_, b_ in [("scan + bg stores + flags", 0, 1), ("prefix", 1, 2), ("winner list", 2, 3), ("table build (gathers)", 3, 4), ("shade", 4, 5)]:
        print("  %-28s mean %7.0f cycles" % (nme, (rs[:, b_] - rs[:, a_]).mean()))

# what makes a strip heavy (round 5, CAD-like batches): least-squares cost per listed face / wide entry / whole-wave entry / candidate
X = np.stack([np.ones_like(tot), d[:, 6], n_wide - n_fat, n_fat, d[:, 7]], 1)
coef, *_ = np.linalg.lstsq(X, tot, rcond=None)
print("cycles ~ %.0f + %.1f per listed face + %.1f per thin wide entry + %.1f per whole-wave entry + %.2f per candidate" % tuple(coef))
print("shares of the summed cycles: base %.1f %%, listed %.1f %%, thin wide %.1f %%, whole-wave %.1f %%, candidates %.1f %%" % tuple(100 * coef * X.sum(0) / tot.sum()))
# round 6: the candidates live in the gap of the strip's list region (two per 16-byte entry): entries a strip needs = listed + candidates / 2
need = d[:, 6] + np.ceil(d[:, 7] / 2)
srt = np.sort(need)[::-1]
print("list-region demand (listed faces + candidates / 2) per strip: max %d, next %s; strips over 2560 / 2752 / 3008: %d / %d / %d" % (
    srt[0], " ".join("%d" % x for x in srt[1:8]), (need > 2560).sum(), (need > 2752).sum(), (need > 3008).sum()))
order = np.argsort(-tot)[:12]
print("heaviest strips: cycles | listed, thin wide, whole-wave, candidates | init, raster(w0), wait, settle+resolve")
for i in order:
    print("  %7.0f | %5.0f %5.0f %5.0f %6.0f | %6.0f %7.0f %7.0f %7.0f" % (tot[i], d[i, 6], n_wide[i] - n_fat[i], n_fat[i], d[i, 7], spans[0][i], spans[1][i], spans[2][i], spans[3][i]))
