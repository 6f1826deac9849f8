"""Debug aid: per-tensor gradient error of one ChangeFormer fp32 training step against the oracle (GPU)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_changeformer_gpu import build, data, loss_fn, oracle_step, DEV
from oracle import changeformer_ref as R
from tests import _util

cfg_name, B, H, W, out_ch = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
drop = tuple(float(v) for v in sys.argv[6].split(",")) if len(sys.argv) > 6 and sys.argv[6] != "-" else None
f64 = len(sys.argv) > 7
ocfg, st, m = build(cfg_name, "fp32", out_ch, drop=drop)
x1, x2, tgt = data(B, H, W, out_ch)
m.train(); m.set_seed(424242)
outs = m(x1.to(DEV), x2.to(DEV))
loss = loss_fn(outs[-1], tgt.to(DEV)); loss.backward(); torch.cuda.synchronize()
masks = R.engine_masks(ocfg, B, H, W, 424242, sites=m._engine.cf_sites(), dtype=torch.float64 if f64 else torch.float32)
if f64:
    st = {k: (v.double() if v.dtype.is_floating_point else v) for k, v in st.items()}
    x1, x2 = x1.double(), x2.double()
R.TAPS = {}
ref, routs, rloss = oracle_step(ocfg, st, x1, x2, tgt if out_ch > 1 or not f64 else tgt.double(), masks)
taps, R.TAPS = R.TAPS, None
ws = m._engine.ws_tensors()
for name, t in taps.items():
    if name in ws:
        e_ = ws[name].float().cpu().double()
        print(f"tap {name:28s} max|diff| {float((e_ - t.double()).abs().max()):.3e}  scale {float(t.abs().max()):.3e}")
for k, (o, r) in enumerate(zip(outs, routs)):
    print("out", k, float((o.detach().cpu().double() - r.detach().double()).abs().max()), float(r.abs().max()))
print("loss", loss.item(), rloss.item())
for name, p in m.named_parameters():
    rg = ref[name].grad
    if rg is None or float(rg.abs().max()) < 1e-9:
        continue
    rel, cos = _util.rel_l2_cos(p.grad.detach().cpu().numpy(), rg.numpy())
    print(f"{rel:.3e} {cos:.7f} {name}")
