"""Full-size parity probe on the device: one [28672 x 8192] K = 2 merge (the Llama-3-70B MLP shape) against the
exact-norm oracle; prints the residual beyond the N largest bins of the difference's spectrum.
    python tools/full70b_check.py"""
import sys, time, torch, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from oracle import spectral_oracle as so
from shardmerge_amd.engine import get_engine
from tests import parity_checks as pc
torch.set_num_threads(16)
eng = get_engine("cuda")
rows, cols = 28672, 8192
t0 = time.time()
base, fts = so.synthetic_layer(rows, cols, 2, seed=4242)
print("inputs", time.time() - t0, flush=True)
t0 = time.time()
trx = so.LayerTrace()
with so.exact_norms():
    refx = so.merge_layer(fts, [base, base], so.ALPHAS[:2], base, trace=trx)
print("oracle K=2", time.time() - t0, flush=True)
out, rep, delta = eng.merge_layer(fts, [base, base], so.ALPHAS[:2], base, want_delta=True)
pc.check_layer_steps(rep, trx, out.numel())
t0 = time.time()
d_total, d_resid = pc.spectral_residual(delta.cpu(), trx.merged_delta, drop=64)
print("residual calc", time.time() - t0)
for drop in (64, 256, 1024, 4096, 16384):
    print("  drop", drop, "-> %.3e" % pc.spectral_residual(delta.cpu(), trx.merged_delta, drop=drop)[1], flush=True)
print("delta total %.3e beyond-ties %.3e out %.3e mism %.4f" % (d_total, d_resid, so.rel_err(out.cpu().float(), refx.float()),
      (out.cpu().view(torch.int16) != refx.view(torch.int16)).float().mean().item()))
import resource
print("maxrss GB", resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6)
