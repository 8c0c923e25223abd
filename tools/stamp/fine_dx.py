import sys, os
sys.argv = [sys.argv[0]]
exec(open("tools/wg_timeline.py").read().split("names = [")[0])
a = np.frombuffer(buf, dtype=np.int64).reshape(K, W, 8).astype(np.float64)
x, y = a[3], a[6]
sel = x[:, 0] > 0
t0 = x[sel, 0]
m = lambda v: np.median(v[sel] - t0) * 0.01
print("conv3_dx (us after workgroup start): dY loads issued %.2f  W loads issued %.2f  all landed (thread 0) %.2f  W stored %.2f  dY committed %.2f  barrier %.2f  mfma done %.2f  end %.2f"
      % (m(y[:, 0]), m(y[:, 1]), m(y[:, 2]), m(y[:, 3]), m(y[:, 4]), m(x[:, 3]), m(x[:, 4]), m(x[:, 6])))
