import sys, torch
sys.path.insert(0, "/root/repo")
from oracle import cpu_step as O
from tests.golden_io import Golden, initial_params
from tests.gpu_util import disc_params, disc_param_names
from gan_image_captioning_amd import engine as E
dev = torch.device("cuda:0")
g = Golden("cfg1"); _, dp = initial_params(g); m = g.meta
eng = E.DiscEngine(m["V"], m["De"], m["R"], m["fs"], m["nf"], 0)
names = disc_param_names(3); params = disc_params(dp, dev); masks = g.masks(0)
rg = torch.Generator().manual_seed(3)
B, Lc, V = m["B"], m["L"], m["V"]
soft = torch.softmax(torch.randn(B, Lc, V, generator=rg) * 2, -1)
caps = g.t("caps")
dl_r = torch.randn(B * m["R"], generator=rg); dl_f = torch.randn(B * m["R"], generator=rg)
for which in ("real", "soft"):
    leaf = {k: dp[k].clone().requires_grad_(True) for k in names}
    if which == "real":
        inp = torch.nn.functional.one_hot(caps, V).float(); mk = masks[0]; dl = dl_r
        lg, st = eng.fwd(params, None, caps.to(dev), True, mk.to(dev))
        grads, _ = eng.bwd(params, st, None, caps.to(dev), True, dl.to(dev), True, False)
    else:
        inp = soft; mk = masks[1]; dl = dl_f
        sd = eng.soft_input(soft.to(dev))
        lg, st = eng.fwd(params, sd, None, True, mk.to(dev))
        grads, _ = eng.bwd(params, st, sd, None, True, dl.to(dev), True, False)
    o, stages = O.disc_forward(leaf, inp, mk, m["R"], return_stages=True)
    (o * dl).sum().backward()
    torch.cuda.synchronize()
    print("==", which)
    for n, gt in zip(names, grads):
        w = leaf[n].grad; d = (gt.cpu() - w).abs()
        print(f"{n:22s} max abs diff {d.max():.3e}  max |want| {w.abs().max():.3e}  argmax {tuple(int(i) for i in torch.nonzero(d == d.max())[0])}")
    # pooled / argmax agreement
    pooled = st["pooled"][:, :eng.F].cpu(); want_p = stages["pooled"]
    print("pooled max diff", float((pooled - want_p).abs().max()), "gate mismatches", int(((pooled > 0) != (want_p > 0)).sum()))
    # oracle argmax per conv
    emb = stages["emb"].detach().reshape(B, Lc, m["R"], 1)
    off = 0
    for k in range(3):
        w = dp[f"convs.{k}.weight"]; f = w.shape[2]
        win = emb.unfold(1, f, 1)
        con = torch.relu(torch.einsum("btrej,cje->bctr", win, w[:, 0]) + dp[f"convs.{k}.bias"][None, :, None, None])
        am = con.max(dim=2)[1].permute(0, 2, 1).reshape(-1, w.shape[0])     # [B*R, n]
        mine = st["argmax"][:, off:off + w.shape[0]].cpu().long()
        live = want_p[:, off:off + w.shape[0]] > 0
        mism = ((am != mine) & live)
        print(f"conv{k}: argmax mismatches among live entries: {int(mism.sum())}")
        if mism.any():
            idx = torch.nonzero(mism)[:5]
            for r_, c_ in idx.tolist():
                b_, rr = divmod(r_, m["R"])
                print("   row", r_, "ch", c_, "mine", int(mine[r_, c_]), "oracle", int(am[r_, c_]), "vals", con[b_, c_, :, rr].tolist())
        off += w.shape[0]
