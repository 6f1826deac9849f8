"""(test infrastructure) bf16 vs fp32 ChangeFormer engines under the multi-scale loss: per-tensor gradient cosine."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_changeformer_gpu import build, data, multi_scale_loss, loss_fn, DEV

for which in ("cp", "ms"):
    res = {}
    for dtype in ("fp32", "bf16"):
        _, _, m = build("tiny", dtype)
        x1, x2, tgt = data(2, 64, 64, 2)
        m.train(); m.set_multi_scale_train(which == "ms"); m.set_seed(9)
        outs = m(x1.to(DEV), x2.to(DEV))
        (multi_scale_loss(outs, tgt.to(DEV)) if which == "ms" else loss_fn(outs[-1], tgt.to(DEV))).backward()
        res[dtype] = {n: p.grad.detach().float().cpu() for n, p in m.named_parameters()}
        res[dtype + "_out"] = [o.detach().float().cpu() for o in outs]
    print("==", which, "outputs rel err:", [float((a - b).norm() / (a.norm() + 1e-12)) for a, b in zip(res["fp32_out"], res["bf16_out"])])
    rows = []
    for n, g32 in res["fp32"].items():
        g16 = res["bf16"][n]
        if float(g32.abs().max()) < 1e-7 or g32.numel() < 2:
            continue
        c = float(torch.nn.functional.cosine_similarity(g32.flatten().double(), g16.flatten().double(), dim=0))
        rows.append((c, n, float(g16.norm() / (g32.norm() + 1e-30))))
    rows.sort()
    for c, n, r in rows[:25]:
        print(f"  {c:8.4f}  norm ratio {r:8.3f}  {n}")
    print("  ... median", rows[len(rows) // 2][0], "n", len(rows))
    if which == "ms":
        for n in res["fp32"]:
            if "make_pred" in n and res["fp32"][n].numel() <= 8:
                print(n, res["fp32"][n].flatten().tolist(), res["bf16"][n].flatten().tolist())
