"""Helpers to read tests/golden/*.npz (written by oracle/make_golden.py)."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SAMPLE_STRIDE = 257


class Golden:
    def __init__(self, name):
        self.z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
        self.meta = json.loads(str(self.z["meta"])) if "meta" in self.z.files else {}

    def t(self, key):
        return torch.from_numpy(np.asarray(self.z[key]))

    def has(self, key):
        return key in self.z.files

    def group(self, prefix):
        return {k[len(prefix):]: self.t(k) for k in self.z.files if k.startswith(prefix)}

    def masks(self, step):
        m = self.meta
        feat = sum(m["nf"])
        packed = self.z[f"s{step}/masks"]
        bits = np.unpackbits(packed, axis=-1)[..., :feat]
        return [torch.from_numpy(bits[i].astype(np.float32)) for i in range(3)]

    def us(self, step):
        u = self.t(f"s{step}/u")
        return [u[i] for i in range(u.shape[0])]


def summarize(t):
    f = t.detach().double().reshape(-1).cpu()
    head = torch.stack([f.sum(), f.abs().sum(), f.pow(2).sum().sqrt()])
    return torch.cat([head, f[::SAMPLE_STRIDE]])


def initial_params(g):
    """(gp, dp) for a fixture: shipped weights, or the seeded recipe + checksum."""
    from oracle import cpu_step as O
    m = g.meta
    if m.get("full", True):
        return g.group("gp0/"), g.group("dp0/")
    pg = torch.Generator().manual_seed(m["param_seed"])
    gp = O.make_gen_params(m["V"], m["E"], m["H"], m["NL"], pg, trunk_feat_dim=m.get("trunk_feat_dim"))
    dp = O.make_disc_params(m["V"], pg, embed_dim=m["De"], num_rep=m["R"], filter_sizes=m["fs"], num_filters=m["nf"])
    for k, v in {**gp, **dp}.items():
        got, want = summarize(v), g.t("p0sum/" + k)
        torch.testing.assert_close(got[3:], want[3:], rtol=0, atol=0)          # strided sample: bit-exact
        torch.testing.assert_close(got[:3], want[:3], rtol=1e-10, atol=0)      # fp64 sums: thread-count dependent order
    return gp, dp
