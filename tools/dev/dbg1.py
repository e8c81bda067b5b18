import sys, os, numpy as np, importlib
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
import __graft_entry__ as ge
mm = ge.load_package(); wl = importlib.import_module(mm.__name__ + '.workloads')
o, oc = ge.load_oracle(); import graphs
g = wl.l2r_hmm(3)
cf = mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))
def run(lens, N, kernel=None, val=1.0):
    B = len(lens)
    V = np.full((B, N, g.P), val, dtype=np.float32)
    if kernel:
        os.environ['MM_DEBUG'] = '1'; os.environ['MM_KERNEL'] = kernel
    else:
        os.environ.pop('MM_DEBUG', None); os.environ.pop('MM_KERNEL', None)
    gam, ttl = mm.batch(*([cf] * B)).pdfposteriors(V, np.asarray(lens, dtype=np.int32))
    return gam, ttl
for lens, N in (([5], 7), ([7, 5], 7), ([5, 7], 7), ([5], 5), ([5], 6), ([4], 6), ([3, 4, 5, 6], 7)):
    for k in (None, 'quad'):
        gam, ttl = run(lens, N, k)
        print(lens, N, k, 'nan' if np.isnan(gam).any() else 'ok', ttl, [np.isnan(gam[b]).any() for b in range(len(lens))])
gam, ttl = run([5], 7)
print(gam[0].T)
for val in (0.0, 1.0):
    gam, ttl = run([5], 7, None, val); print(val, gam[0, :, 0])
