"""RB_STAMP build: for the first conv layer's launch, the (start, end) of the workgroups that shared a CU (are they co-resident?)."""
import sys, os
sys.argv = [sys.argv[0]]
exec(open("tools/wg_timeline.py").read().split("names = [")[0])
a = np.frombuffer(buf, dtype=np.int64).reshape(K, W, 8).astype(np.float64)
rows = a[0]
idx = np.nonzero(rows[:, 0] > 0)[0]
r = rows[idx]
t0 = r[:, 0].min()
hw = r[:, 7].astype(np.int64)
cu = (hw >> 16) * 4096 + ((hw >> 8) & 0xff)
simd = (hw >> 4) & 3
shown = 0
for c in np.unique(cu):
    sel = np.nonzero(cu == c)[0]
    if len(sel) < 2:
        continue
    print("CU %6d:" % c, "  ".join("wg %4d [%.2f .. %.2f] hw %04x" % (idx[i], (r[i, 0] - t0) * 0.01, (r[i, 6] - t0) * 0.01, hw[i] & 0xffff) for i in sel))
    shown += 1
    if shown >= 6:
        break
late = (r[:, 0] - t0) * 0.01 > 3
print("late starters: %d of %d; their wg index mod 8:" % (late.sum(), len(r)), np.bincount(idx[late] % 8, minlength=8).tolist())
