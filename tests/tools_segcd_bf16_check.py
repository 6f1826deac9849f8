"""bf16 SegCD against the fp32 engine on the same inputs (forward error, per-tensor gradient rel-l2 / cosine) under several
kernel-selection switches.  Run on the GPU box: python tests/tools_segcd_bf16_check.py"""
import sys, os, subprocess, pickle
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

def run(dt, out):
    import torch
    from oracle import segcd_ref as G
    from stcd_amd.segcd import SegCD
    from stcd_amd.losses import bce_dice_with_logits
    dev = "cuda:0"
    B, H, W = 4, 128, 128
    rng = np.random.default_rng(5)
    x1 = torch.from_numpy(rng.standard_normal((B, 3, H, W)).astype(np.float32)).to(dev)
    x2 = torch.from_numpy(rng.standard_normal((B, 3, H, W)).astype(np.float32)).to(dev)
    tgt = torch.from_numpy((rng.random((B, 1, H, W)) < 0.2).astype(np.float32)).to(dev)
    st = G.synth_state(3, 1, 9, perturb_running=True)
    rg = float(os.environ.get("SEGCD_RES_GAMMA", "1"))
    for k in st:
        if k.endswith("bn3.weight"):
            st[k] = st[k] * rg
    m = SegCD(encoder_name="resnet50", dtype=dt)
    m.load_state_dict(st)
    m.to(dev).eval()
    with torch.no_grad():
        ev = [t.cpu().double().numpy() for t in m(x1, x2)]
    m.train()
    o = m(x1, x2)
    loss = bce_dice_with_logits(o[2], tgt) + bce_dice_with_logits(o[0], tgt) + 0.5 * o[1].mean()
    loss.backward()
    torch.cuda.synchronize()
    pickle.dump((ev, [t.detach().cpu().double().numpy() for t in o], loss.item(),
                 {n: p.grad.cpu().double().numpy().ravel() for n, p in m.named_parameters()}), open(out, "wb"))

def cmp(a, b, tag):
    for k in range(3):
        print(tag, "eval out", k, "rel-l2 %.3e" % (np.linalg.norm(a[0][k] - b[0][k]) / np.linalg.norm(a[0][k])),
              "train rel-l2 %.3e" % (np.linalg.norm(a[1][k] - b[1][k]) / np.linalg.norm(a[1][k])))
    rows = []
    for n in a[3]:
        x, y = a[3][n], b[3][n]
        nx = np.linalg.norm(x)
        if nx < 1e-12:
            continue
        rows.append((np.linalg.norm(x - y) / nx, float(x @ y) / (nx * np.linalg.norm(y) + 1e-30), n))
    rows.sort()
    print(tag, "grad rel-l2 median %.3e" % rows[len(rows) // 2][0], "worst %.3e %s" % (rows[-1][0], rows[-1][2]), "min cos %.4f" % min(r[1] for r in rows))
    if "-v" in sys.argv:
        for n in list(a[3])[::-1][::6]:
            x, y = a[3][n], b[3][n]
            nx = np.linalg.norm(x)
            if nx > 1e-12:
                print("    %-44s rel %.3f cos %.4f norm ratio %.3f" % (n, np.linalg.norm(x - y) / nx, float(x @ y) / (nx * np.linalg.norm(y) + 1e-30), np.linalg.norm(y) / nx))

if len(sys.argv) > 2 and sys.argv[1] == "run":
    run(sys.argv[2], sys.argv[3])
    sys.exit(0)
os.makedirs("gpurun_out", exist_ok=True)
for rg in ("1",):
    variants = [("fp32", "fp32", {}), ("bf16", "bf16", {})]
    res = {}
    for name, dt, env in variants:
        f = f"/tmp/segcd_{name}.pkl"
        subprocess.run([sys.executable, __file__, "run", dt, f], env=dict(os.environ, SEGCD_RES_GAMMA=rg, **env), check=True)
        res[name] = pickle.load(open(f, "rb"))
        if name != "fp32":
            cmp(res["fp32"], res[name], f"res_gamma={rg} {name}")
