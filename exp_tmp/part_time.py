import sys, torch
sys.path.insert(0, "/root/repo")
from shardmerge_amd.engine import get_engine
eng = get_engine("cuda:0")
torch.manual_seed(0)
for kind, n in (("bf16", 8192 * 8192), ("f32", 8192 * 8192)):
    if kind == "bf16":
        b = (torch.randn(n, device="cuda") * 0.02).bfloat16(); x = (b.float() + torch.randn(n, device="cuda") * 0.003).bfloat16()
    else:
        b = None; x = torch.randn(n, device="cuda") * 0.003
    eng.reference_cpu_norm(x, b)
    eng.ctx.profile(True); eng.ctx.profile_reset()
    for _ in range(3): eng.reference_cpu_norm(x, b)
    t = eng.ctx.profile_table(); eng.ctx.profile(False)
    print(kind, {k: round(v[1] / v[0] * 1e3, 1) for k, v in t.items()})
