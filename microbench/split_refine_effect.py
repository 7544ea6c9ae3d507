import sys, warnings
sys.path.insert(0, '.')
import numpy as np
from oracle import plspy_oracle as orc
from plspy_amd import split_half_resampling as sh
rs = np.random.RandomState(7)
X = rs.randn(120, 200_000)
Y = rs.randn(120, 1) + rs.randn(120, 8) * np.logspace(0, -3.5, 8)[None, :]
co = np.array([[20] * 3, [20] * 3]); bscan = [1, 2]
mask = orc.bscan_mask(co, bscan)
kw = dict(mctype=0, bscan=bscan, Xbscan=X[mask], Ybscan=Y[mask])
S = 2
with warnings.catch_warnings(), np.errstate(all="ignore"):
    warnings.simplefilter("ignore")
    np.random.seed(3)
    ott = orc.split_half_both("mb", X, Y, co, S, which="tt", mctype=0, bscan=bscan, Ybscan=Y[mask], lv=2, only={0})
for ratio in (1e-5, 0.0):
    sh.REFINE_RATIO = ratio
    np.random.seed(3)
    tt = sh.split_half_test_train("mb", X, Y, co, S, **kw)
    got, want = tt["pls_s_train"][0, :36, 0], ott["pls_s_train"][0, :36, 0]
    print("REFINE_RATIO", ratio, "max rel err of s:", np.max(np.abs(got - want) / want), "s range", want[0], want[35])
