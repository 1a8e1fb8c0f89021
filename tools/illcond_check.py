import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from robo_amd import _lib
from oracle import gp_oracle as O
ctx = _lib.Context(0)
rs = np.random.RandomState(0)
for N, D, noise, ls in ((1500, 2, 1e-6, 0.3), (1500, 2, 1e-8, 0.5), (3000, 3, 1e-4, 0.5), (1000, 1, 1e-6, 0.2)):
    X = rs.rand(N, D); y = np.sin(4 * X.sum(1))
    theta = np.concatenate([[0.0], np.full(D, np.log(ls ** 2)), [np.log(noise)]])
    g = _lib.DeviceGP(ctx, "matern52", N, D); g.set_data(X, y)
    c = float(y.mean())
    try:
        ll = g.fit(theta, c)
    except np.linalg.LinAlgError as e:
        print(N, D, noise, "device: not PD"); 
        try:
            O.gp_compute("matern52", theta, X); print("  oracle: PD!")
        except np.linalg.LinAlgError: print("  oracle: not PD either")
        continue
    L = O.gp_compute("matern52", theta, X)
    llo = O.gp_log_likelihood(L, y, c)
    Xs = rs.rand(2000, D)
    mu, var = g.predict(Xs)
    muo, varo = O.gp_predict_diag("matern52", theta, L, X, y, c, Xs)
    K = O.kernel_matrix("matern52", theta[:-1], X); K[np.diag_indices(N)] += noise
    ev = np.linalg.eigvalsh(K); cond = ev[-1] / ev[0]
    print(N, D, noise, "cond %.2e" % cond, "ll rel %.2e" % abs((ll - llo) / llo), "mu abs %.2e" % np.max(np.abs(mu - muo)),
          "var abs %.2e" % np.max(np.abs(var - varo)), "min var", var.min(), varo.min())
