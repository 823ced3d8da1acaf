"""Loading of the committed golden fixtures (tests/golden/*.npz) for the test-suite."""
import os

import numpy as np
import torch

from oracle import rgl_oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_cache = {}


def load(name):
    if name not in _cache:
        _cache[name] = dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))
    return _cache[name]


def master(flavour):
    return load("weights_" + flavour)


def graph_sd(m, which, L, similarity="embedded_gaussian"):
    """Reference-style RGL state-dict (torch tensors) cut out of a master weight set."""
    pre = which + "."
    sd = {k[len(pre):]: torch.tensor(v) for k, v in m.items()
          if k.startswith(pre) and not k[len(pre):].startswith("Ws.")}
    for l in range(L):
        sd["Ws.%d" % l] = torch.tensor(m[pre + "Ws.%d" % l])
    if similarity == "concatenation":
        sd.pop("w_a")
        for k, v in m.items():
            if k.startswith("concat_w_a."):
                sd["w_a." + k[len("concat_w_a."):]] = torch.tensor(v)
    elif similarity != "embedded_gaussian":
        sd.pop("w_a")
    return sd


def sub_sd(m, which):
    pre = which + "."
    return {k[len(pre):]: torch.tensor(v) for k, v in m.items() if k.startswith(pre)}


def checkpoint(flavour, L=2, variant="separate", similarity="embedded_gaussian"):
    """The nested dict ModelPredictiveRL.get_state_dict() would return for these weights."""
    m = master(flavour)
    if variant == "separate":
        return {"graph_model1": graph_sd(m, "graph_model1", L, similarity),
                "graph_model2": graph_sd(m, "graph_model2", L, similarity),
                "value_network": sub_sd(m, "value_network"),
                "motion_predictor": sub_sd(m, "motion_predictor")}
    if variant == "shared":
        return {"graph_model": graph_sd(m, "graph_model1", L, similarity),
                "value_network": sub_sd(m, "value_network"),
                "motion_predictor": sub_sd(m, "motion_predictor")}
    if variant == "linear":
        return {"graph_model": graph_sd(m, "graph_model1", L, similarity),
                "value_network": sub_sd(m, "value_network")}
    raise KeyError(variant)


def oracle_params(flavour, L=2, variant="separate", similarity="embedded_gaussian"):
    return orc.MprlParams.from_checkpoint(checkpoint(flavour, L, variant, similarity))


def forward_cases():
    """Forward KATs: file "forward" (round 1: L <= 3) and "forward_l4" (four GCN layers)."""
    out = []
    for name in ("forward", "forward_l4"):
        fw = load(name)
        for ci, line in enumerate(fw["forward_cases"]):
            H, B, L, flavour, sim, lw, sk = str(line).split("|")
            out.append(dict(file=name, idx=ci, H=int(H), B=int(B), L=int(L), flavour=flavour, sim=sim,
                            layerwise=bool(int(lw)), skip=bool(int(sk))))
    return out


def plan_cases():
    pl = load("planning")
    out = []
    for line in pl["plan_cases"]:
        tag, sk, D, w, clip, sparse, variant, flavour = str(line).split("|")
        out.append(dict(tag=tag, scene=sk, D=int(D), w=int(w), clip=bool(int(clip)), sparse=bool(int(sparse)),
                        variant=variant, flavour=flavour))
    return out


def root_clip_cases():
    """Fixture "root_clip" (round 5): what upstream computes INSIDE the root's action_clip, on genuine float64 roots."""
    rc = load("root_clip")
    out = []
    for line in rc["rootclip_cases"]:
        tag, D, w, sparse, variant = str(line).split("|")
        out.append(dict(tag=tag, D=int(D), w=int(w), sparse=bool(int(sparse)), variant=variant))
    return out


def path_g_sd():
    g = load("path_g")
    return {k[len("g.weights."):]: torch.tensor(v) for k, v in g.items() if k.startswith("g.weights.")}
