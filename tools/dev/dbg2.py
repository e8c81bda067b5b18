import sys, os, numpy as np, importlib
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
import __graft_entry__ as ge
mm = ge.load_package(); wl = importlib.import_module(mm.__name__ + '.workloads')
o, oc = ge.load_oracle(); import graphs
def check(g, lens, N, seed=0, scale=1.0):
    B = len(lens)
    rng = np.random.default_rng(seed)
    V = (scale * rng.standard_normal((B, N, g.P))).astype(np.float32)
    lens = np.asarray(lens, dtype=np.int32)
    g_ref, t_ref = oc.batch_shared(graphs.to_oracle(o, g), g.state2pdf, g.P, V, lens, dtype=np.float64)
    cf = mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))
    bf = mm.batch(*([cf] * B))
    gam, ttl = bf.pdfposteriors(V, lens)
    err = np.abs(gam - g_ref).max()
    ok = np.isfinite(t_ref)
    terr = np.abs(ttl[ok] - t_ref[ok]).max() if ok.any() else 0
    print(g.name, lens.tolist(), N, 'gamma err %.2e' % err, 'ttl err %.2e' % terr, 'nan' if np.isnan(gam).any() else '', bf.kernels()[:40])
    return err
check(wl.l2r_hmm(3), [5, 5], 5)
check(wl.l2r_hmm(3), [7, 5], 7)
check(wl.l2r_hmm(3), [1, 1], 3)
check(wl.l2r_hmm(3), [2, 3, 4], 6)
check(wl.random_fsm(40, 6, 3.0, seed=1), [33, 26, 19, 12, 5], 33)
check(wl.lfmmi_denominator(600, 40, seed=5), [25, 18, 7], 25, scale=1.5)
check(wl.lfmmi_denominator(2000, 84), [40, 33, 26], 40)
check(wl.lfmmi_denominator(2000, 84), [300] * 4, 300)
