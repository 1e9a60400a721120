import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cellector_amd import Cellector
L, N, d = [int(x) for x in sys.argv[1:3]] + [float(sys.argv[3])] if len(sys.argv) > 3 else (20000, 20000, 0.01)
res = {}
for eng in (1, 2):
    g = Cellector(0); g.set_option("engine", eng); g.set_option("keep_coo", 0)
    g.load_synthetic(L, N, d, seed=4, minority_fraction=0.05)
    g.em_iteration(5.0)
    res[eng] = g.cell_outputs()
    if eng == 2:
        for rep in range(2):
            g2 = Cellector(0); g2.set_option("keep_coo", 0); g2.load_synthetic(L, N, d, seed=4, minority_fraction=0.05)
            g2.em_iteration(5.0); r2 = g2.cell_outputs()
            print("rerun identical:", np.array_equal(r2["ll"], res[2]["ll"])); g2.close()
    g.close()
diff = np.abs(res[1]["ll"] - res[2]["ll"])
bad = np.nonzero(diff > 1e-7)[0]
print("bad cells:", len(bad), "of", N)
if len(bad):
    print("first bad:", bad[:40])
    print("blocks histogram:", np.bincount(bad // 1024, minlength=(N + 1023) // 1024))
    print("diff sample:", diff[bad[:10]], "ll1:", res[1]["ll"][bad[:5]], "ll2:", res[2]["ll"][bad[:5]])
    print("|ll2|<|ll1| fraction:", np.mean(np.abs(res[2]["ll"][bad]) < np.abs(res[1]["ll"][bad])))
