"""Developer tool (round 6): what a better ORDER or work stealing could buy the CAD-like launch — list-scheduling simulations over the measured
start / end stamps of every workgroup (chip-wide clock):
    MVR_DIAG_DUMP=gpurun_out/wg.npy MVR_DIAG_EXTRA=-DMVR_DIAG_WALL MVR_GEOMETRY=cad python3 tools/diag_fast.py     (on the GPU box)
    python3 tools/r06_sched_sim.py gpurun_out/wg.npy                                                                 (anywhere)
128 workgroup slots per XCD; a workgroup's duration is taken as measured (under the contention of its launch: indicative, not exact)."""
import heapq
import sys

import numpy as np

a = np.load(sys.argv[1])
dur = np.where(a[:, 5] > 0, (a[:, 5] - a[:, 0]) * 0.01, 0.0)   # us
listed = (a[:, 6] & 0xffffffff).astype(float)
wide = (a[:, 6] >> 32).astype(float)
cand = (a[:, 7] & 0xffffffff).astype(float)
fat = (a[:, 7] >> 32).astype(float)


def sim(queues, slots, steal=False):
    rem = [float(np.sum(q)) for q in queues]
    pos = [0] * len(queues)
    heap = [(0.0, q, s) for q in range(len(queues)) for s in range(slots)]
    heapq.heapify(heap)
    end = 0.0
    while heap:
        t, q, s = heapq.heappop(heap)
        src = q
        if pos[q] >= len(queues[q]):
            live = [k for k in range(len(queues)) if pos[k] < len(queues[k])]
            if not steal or not live:
                continue
            src = max(live, key=lambda k: rem[k])
        d = queues[src][pos[src]]
        pos[src] += 1
        rem[src] -= d
        end = max(end, t + d)
        heapq.heappush(heap, (t + d, q, s))
    return end


def per_xcd(w=None):
    out = []
    for x in range(8):
        d = dur[x::8]
        out.append(d if w is None else d[np.argsort(-w[x::8], kind="stable")])
    return out


print("sum of the durations / 1024 slots: %.1f us" % (dur.sum() / 1024))
print("launch order, per-XCD queues: %.1f us; + stealing from the fullest XCD: %.1f" % (sim(per_xcd(), 128), sim(per_xcd(), 128, True)))
print("ONE queue, longest first (true durations): %.1f" % sim([np.sort(dur)[::-1]], 1024))
for name, w in (("true duration", dur), ("listed faces", listed), ("listed + 28 x whole-wave entries", listed + 28 * fat),
                ("... + 2.7 x thin wide entries", listed + 28 * fat + 2.7 * (wide - fat)),
                ("... + 0.8 x candidates", listed + 28 * fat + 2.7 * (wide - fat) + 0.8 * cand)):
    print("per-XCD queues ordered by %-34s %.1f us (+ stealing %.1f)   correlation with the duration %.3f" % (
        name + ":", sim(per_xcd(w), 128), sim(per_xcd(w), 128, True), np.corrcoef(w[dur > 0], dur[dur > 0])[0, 1]))
