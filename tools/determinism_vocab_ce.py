"""Run-to-run determinism of dc_vocab_ce on the case that once failed its tolerance in a full-suite run (M = 130, V = 50 000, K = 1024, bf16,
keras_sparse, fp32 gradient): N identical calls, every output compared bit for bit with the first call's.
Usage: python tools/determinism_vocab_ce.py [M] [calls] [bf16-gradient 0/1]     (DCAP_LIB selects an experiment build of the library)"""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from image_captioning_amd import ops


def dev(a, dtype=torch.float32): return torch.tensor(np.ascontiguousarray(a), dtype=dtype, device="cuda")


M = int(sys.argv[1]) if len(sys.argv) > 1 else 130
CALLS = int(sys.argv[2]) if len(sys.argv) > 2 else 400
BF16_DL = bool(int(sys.argv[3])) if len(sys.argv) > 3 else False
V, K = 50000, 1024
rng = np.random.default_rng(V + K)
X = rng.standard_normal((M, K)); W = rng.standard_normal((K, V)) * (2.0 / np.sqrt(K)); b = rng.standard_normal(V); t = rng.integers(0, V, M)
b[t[0]] -= 80.0; X[1] = 0
w = rng.random(M); w[2] = 0.0
Xd, Wd = ops.to_bf16(dev(X)), ops.to_bf16(dev(W))
bd, td, wd = dev(b), dev(t, torch.int32), dev(w)
ldd = (V + 7) // 8 * 8
ref = None
bad = 0
for it in range(CALLS):
    loss = torch.empty(M, device="cuda"); db = torch.empty(V, device="cuda")
    dl = torch.empty((M, ldd), device="cuda", dtype=torch.bfloat16) if BF16_DL else torch.empty((M, V), device="cuda")
    ops.vocab_ce(Xd, Wd, bd, td, loss_rows=loss, dlogits=dl, dbias=db, grad_scale=1.0, row_weights=wd, keras_sparse=True)
    torch.cuda.synchronize()
    if ref is None:
        ref = (loss.clone(), dl.clone(), db.clone())
        continue
    for name, a, r in (("loss", loss, ref[0]), ("dl", dl, ref[1]), ("db", db, ref[2])):
        if torch.equal(a, r):
            continue
        bad += 1
        a, r = a.float(), r.float()
        nz = torch.nonzero(a != r).cpu().numpy()
        print("call %d: %s differs from the first call's in %d entries" % (it, name, len(nz)), flush=True)
        if name == "dl":
            for i, j in nz[:32]:
                print("    row %4d col %6d (tile col %3d, lane pair c4 %2d, component %d): %.9g  first call %.9g  diff %.3e   bias %.6f"
                      % (i, j, j % 128, (j % 128) // 4, j % 4, float(a[i, j]), float(r[i, j]), float(a[i, j] - r[i, j]), b[j]), flush=True)
print("done: %d differing outputs in %d calls (M = %d, %s gradient)" % (bad, CALLS, M, "bf16" if BF16_DL else "fp32"))
