import sys, os
sys.argv = [sys.argv[0]]
exec(open("tools/wg_timeline.py").read().split("names = [")[0])
a = np.frombuffer(buf, dtype=np.int64).reshape(K, W, 8).astype(np.float64)
rows = a[0]
idx = np.nonzero(rows[:, 0] > 0)[0]
r = rows[idx]
img = idx // 16
first = img < 51
for nm, sel in (("first", first), ("second", ~first)):
    x = r[sel]
    y = a[4][idx][sel]
    t0 = x[:, 0]
    m = lambda v: np.median(v - t0) * 0.01
    print("%s (us after workgroup start): W issued %.2f  X issued %.2f  landed(thread 0) %.2f  LDS stores issued %.2f  barrier passed %.2f  mfma done %.2f  end %.2f"
          % (nm, m(x[:, 1]), m(y[:, 0]), m(y[:, 1]), m(y[:, 2]), m(x[:, 3]), m(x[:, 4]), m(x[:, 6])))
