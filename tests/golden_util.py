"""Loading of the committed golden fixtures (tests/golden/*.npz)."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))

from cases import CASES, FIXED, build_data, build_nets  # noqa: E402,F401


def names():
    return sorted(CASES)


def load(name):
    z = np.load(os.path.join(HERE, "golden", name + ".npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    fx = {"meta": meta, "X": z["X"], "y": z["y"] if "y" in z.files else None, "fixed_noise": z["fixed_noise"],
          "samples": z["samples"], "losses": z["losses"], "present": z["losses_present"]}
    fx["X_test"] = z["X_test"] if "X_test" in z.files else None
    fx["y_test"] = z["y_test"] if "y_test" in z.files else None
    fx["logged"] = None
    for k in z.files:
        if k.startswith("logged/"):
            _, mode, key = k.split("/", 2)
            fx["logged"] = fx["logged"] or {}
            fx["logged"].setdefault(mode, {})[key] = z[k]
    fx["Z"] = [z["Z/%d" % i] for i in range(len([k for k in z.files if k.startswith("Z/")]))]
    fx["alpha"] = [z["alpha/%d" % i] for i in range(len([k for k in z.files if k.startswith("alpha/")]))]
    for phase in ("init", "final"):
        fx[phase] = {}
        for k in z.files:
            if k.startswith(phase + "/"):
                _, net, key = k.split("/", 2)
                fx[phase].setdefault(net, {})[key] = z[k]
    return fx


def loss_rows(fx):
    """[(who, {name: value})] for every recorded sub-step."""
    keys, who = fx["meta"]["loss_keys"], fx["meta"]["who"]
    rows = []
    for i, w in enumerate(who):
        rows.append((w, {k: fx["losses"][i, j] for j, k in enumerate(keys) if fx["present"][i, j]}))
    return rows


OUTLIER_FRACTION = 0.001   # share of a tensor's elements allowed outside `rel` (at least one element); the largest
                           # share any large tensor of any case uses is 0.09% (the run's summary prints the usage)


def shadowed_keys(network):
    """state_dict keys that cannot be compared element-wise: the bias of a layer whose output goes straight into a
    BatchNorm, and that BatchNorm's running_mean.  BatchNorm subtracts the batch mean, so such a bias has a
    mathematically zero gradient; what the optimizer sees is fp32 rounding noise (in the reference as well), which
    RMSprop/Adam normalise into full-size steps: the bias random-walks and the running mean that follows it shifts by
    the same amount.  Neither influences any output of the network.  A bias that reaches the BatchNorm through an
    activation only has its running_mean skipped (the mean then depends on the bias nonlinearly)."""
    from torch import nn
    keys, last, direct = set(), None, False
    for name, m in network.named_modules():
        if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d, nn.Linear)):
            last, direct = (name if m.bias is not None else None), True
        elif isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d)):
            if last is not None:
                keys.add(name + ".running_mean")
                if direct:
                    keys.add(last + ".bias")
            last, direct = (name if m.bias is not None else None), False
        elif not isinstance(m, (nn.Sequential, nn.Identity)) and name:
            direct = False
    return keys


OUTLIER_REPORT = []     # (label, key, outliers, elements) of every tensor compare_network has looked at: printed by conftest
LARGEST_DRIFT = [0.0, ""]   # the largest single-element difference any comparison saw, and where: printed by conftest


def compare_network(got_state, want_state, network, rel, drift, label):
    """Element-wise parameter parity: in every tensor at least 1 - OUTLIER_FRACTION of the elements must agree to
    `rel` relative to the tensor's largest magnitude, and every element must stay within `drift` of the reference.

    Why an outlier budget: RMSprop and Adam divide the gradient by its running magnitude, so an element whose
    gradient is at the fp32 rounding-noise level (any two summation orders disagree on its sign) still takes a
    full learning-rate-sized step, in the reference exactly as here.  From a zero optimizer state the first RMSprop
    step moves EVERY element by 10*lr*sign(g); a handful of sign flips per million elements is the expected
    outcome of comparing two correct fp32 implementations, and they stay bounded by `drift`."""
    skip = shadowed_keys(network)
    problems = []
    for k, want in want_state.items():
        got = got_state[k].detach().cpu().numpy().astype(np.float64)
        want = np.asarray(want, dtype=np.float64)
        if k.endswith("num_batches_tracked"):
            if int(got) != int(want):
                problems.append("%s %s: %d != %d" % (label, k, int(got), int(want)))
            continue
        if got.shape != want.shape:
            problems.append("%s %s: shape %s != %s" % (label, k, got.shape, want.shape))
            continue
        if k in skip:
            continue
        d = np.abs(got - want)
        scale = max(float(np.abs(want).max()), 1e-3)
        outliers = int((d > rel * scale).sum())
        OUTLIER_REPORT.append((label, k, outliers, d.size))
        allowed = max(1, int(OUTLIER_FRACTION * d.size))
        if outliers > allowed:
            problems.append("%s %s: %d of %d elements differ by more than %.0e * %.3e (max %.3e), allowed %d" % (
                label, k, outliers, d.size, rel, scale, d.max(), allowed))
        if d.max() > LARGEST_DRIFT[0]:
            LARGEST_DRIFT[0], LARGEST_DRIFT[1] = float(d.max()), "%s %s" % (label, k)
        if d.max() > drift:
            problems.append("%s %s: an element drifted %.3e > %.1e" % (label, k, d.max(), drift))
    return problems
