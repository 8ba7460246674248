"""Generates the golden fixtures under tests/golden/*.npz.  Run by hand in the build container:

    python tests/golden/make_goldens.py [case ...]

For every case it (1) runs the REFERENCE implementation imported from /root/reference (oracle/ref_import.py) on
seeded inputs, recording the latent batches, the numpy alphas, every sub-step's losses and the final state; (2) runs
the CPU restatement (oracle/restated.py) from the same seeds WITHOUT injecting anything and requires bit-identical
losses, parameters, buffers and samples (this pins the oracle, including its RNG consumption order); (3) writes
inputs + expected outputs as a fixture.  Fixtures hold data only.
"""
import copy
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import bench_configs as cfg  # noqa: E402
from oracle import ref_import  # noqa: E402
from oracle.restated import OracleGAN  # noqa: E402

from cases import CASES, FIXED, build_data as _data, build_nets as _nets  # noqa: E402


def _state(net):
    return {k: v.detach().clone().numpy() for k, v in net.state_dict().items()}


def _ctor_kwargs(case, extra):
    kw = dict(x_dim=case["x_dim"], z_dim=case["z_dim"], optim_kwargs=case.get("optim_kwargs"),
              fixed_noise_size=FIXED)
    if "y_dim" in case:
        kw["y_dim"] = case["y_dim"]
    kw.update(extra)
    return kw


def run_reference(name, case):
    ref = ref_import.load()
    G, D, E = _nets(case)
    init = {"Generator": _state(G), "Adversariat": _state(D)}
    if case["algo"] == "lrgan":
        init["Encoder"] = _state(E)
    X, y = _data(case)
    Xt, yt = _data(case, test=True) if "n_test" in case else (None, None)
    torch.manual_seed(7)
    np.random.seed(7)
    extra = dict(device="cpu", folder=None)
    if case["algo"] == "lrgan":
        extra["encoder"] = E
    model = getattr(ref, case["cls"])(generator=G, adversariat=D, **_ctor_kwargs(case, extra))
    rec = {"Z": [], "alpha": [], "losses": []}
    recording = {"on": False}
    orig_sample = model.sample

    def sample(n):
        z = orig_sample(n)
        if recording["on"]:
            rec["Z"].append(z.detach().clone().numpy())
        return z

    model.sample = sample
    orig_random = np.random.random

    def random(size=None):
        a = orig_random(size)
        if recording["on"]:
            rec["alpha"].append(np.array(a, copy=True))
        return a

    np.random.random = random
    orig_backward = model._backward

    def backward(who=None):
        rec["losses"].append((who, {k: float(v.detach()) for k, v in model._losses.items()}))
        return orig_backward(who=who)

    model._backward = backward
    recording["on"] = True
    fit_kw = dict(epochs=1, batch_size=case["batch"], steps=copy.deepcopy(case.get("steps")), print_every=None,
                  save_losses_every=None, enable_tensorboard=False)
    fit_kw.update(case.get("fit", {}))
    try:
        if y is None:
            model.fit(X, **fit_kw)
        elif Xt is None:
            model.fit(X, y, **fit_kw)
        else:
            model.fit(X, y, X_test=Xt, y_test=yt, **fit_kw)
    finally:
        np.random.random = orig_random
        recording["on"] = False
    nets = {"Generator": G, "Adversariat": D}
    if case["algo"] == "lrgan":
        nets["Encoder"] = E
    final = {k: _state(v) for k, v in nets.items()}
    if y is None:
        samples = model.generate(z=model.fixed_noise)
    else:
        samples = model.generate(y=model.fixed_labels, z=model.fixed_noise)
    return dict(init=init, X=X, y=y, Xt=Xt, yt=yt, Z=rec["Z"], alpha=rec["alpha"], losses=rec["losses"], final=final,
                samples=np.asarray(samples), fixed_noise=model.fixed_noise.detach().numpy().copy(),
                logged=copy.deepcopy(getattr(model, "logged_losses", None)))


def run_oracle(case, inject=None):
    """Restatement run; with inject=None it draws Z/alpha itself from the same seeds as the reference run."""
    G, D, E = _nets(case)
    X, y = _data(case)
    Xt, yt = _data(case, test=True) if "n_test" in case else (None, None)
    torch.manual_seed(7)
    np.random.seed(7)
    o = OracleGAN(case["algo"], G, D, case["x_dim"], case["z_dim"], y_dim=case.get("y_dim"), encoder=E,
                  optim_kwargs=case.get("optim_kwargs"), fixed_noise_size=FIXED)
    kw = {}
    if inject is not None:
        kw = dict(latents=[torch.from_numpy(z) for z in inject["Z"]], alphas=list(inject["alpha"]))
    kw.update(case.get("fit", {}))
    hist = o.fit(X, y, epochs=1, batch_size=case["batch"], steps=copy.deepcopy(case.get("steps")), X_test=Xt, y_test=yt,
                 **kw)
    final = {k: _state(v) for k, v in o.nets.items()}
    ylab = torch.from_numpy(y[:FIXED]) if y is not None else None
    return dict(losses=hist, final=final, samples=o.fixed_samples(ylab), fixed_noise=o.fixed_noise.detach().numpy(),
                logged=o.logged_losses)


def _assert_identical(ref, orc, name):
    assert len(ref["losses"]) == len(orc["losses"]), name
    for (w1, l1), (w2, l2) in zip(ref["losses"], orc["losses"]):
        assert w1 == w2 and l1.keys() == l2.keys(), (name, w1, w2)
        for k in l1:
            a, b = l1[k], l2[k]
            assert a == b or (np.isnan(a) and np.isnan(b)), (name, w1, k, a, b)
    for net in ref["final"]:
        for k, v in ref["final"][net].items():
            assert np.array_equal(v, orc["final"][net][k], equal_nan=True), (name, net, k)
    assert np.array_equal(ref["fixed_noise"], orc["fixed_noise"]), name
    assert np.array_equal(ref["samples"], orc["samples"]), name
    assert (ref["logged"] is None) == (orc["logged"] is None), name
    if ref["logged"] is not None:
        assert ref["logged"].keys() == orc["logged"].keys(), (name, ref["logged"].keys(), orc["logged"].keys())
        for mode in ref["logged"]:
            assert ref["logged"][mode].keys() == orc["logged"][mode].keys(), (name, mode)
            for k, v in ref["logged"][mode].items():
                assert np.array_equal(np.array(v), np.array(orc["logged"][mode][k]), equal_nan=True), (name, mode, k)


def write_fixture(name, case, ref):
    arrays = {"X": ref["X"], "fixed_noise": ref["fixed_noise"], "samples": ref["samples"]}
    if ref["y"] is not None:
        arrays["y"] = ref["y"]
    if ref["Xt"] is not None:
        arrays["X_test"], arrays["y_test"] = ref["Xt"], ref["yt"]
    if ref["logged"] is not None:
        for mode, per_mode in ref["logged"].items():
            for k, v in per_mode.items():
                arrays["logged/%s/%s" % (mode, k)] = np.asarray(v, dtype=np.float64)
    for i, z in enumerate(ref["Z"]):
        arrays["Z/%d" % i] = z
    for i, a in enumerate(ref["alpha"]):
        arrays["alpha/%d" % i] = a
    for phase in ("init", "final"):
        for net, sd in ref[phase].items():
            for k, v in sd.items():
                arrays["%s/%s/%s" % (phase, net, k)] = v
    keys = sorted({k for _, l in ref["losses"] for k in l})
    table = np.full((len(ref["losses"]), len(keys)), np.nan, dtype=np.float64)
    present = np.zeros(table.shape, dtype=bool)
    for i, (_, l) in enumerate(ref["losses"]):
        for k, v in l.items():
            table[i, keys.index(k)] = v
            present[i, keys.index(k)] = True
    arrays["losses"] = table
    arrays["losses_present"] = present
    meta = {k: v for k, v in case.items()}
    meta.update(name=name, loss_keys=keys, who=[w for w, _ in ref["losses"]], torch=torch.__version__,
                fixed_noise_size=FIXED, seed=7,
                pinned="reference and oracle/restated.py bit-identical at generation time")
    arrays["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    return path


def main(argv):
    names = argv or list(CASES)
    torch.set_num_threads(1)  # one thread: the summation order of the CPU kernels does not depend on the host
    for name in names:
        case = CASES[name]
        ref = run_reference(name, case)
        orc = run_oracle(case)
        _assert_identical(ref, orc, name)
        orc2 = run_oracle(case, inject=ref)
        _assert_identical(ref, orc2, name + " (injected)")
        path = write_fixture(name, case, ref)
        print("%-20s %3d sub-steps  reference == oracle (bitwise)  -> %s (%.0f KB)" % (
            name, len(ref["losses"]), os.path.relpath(path, ROOT), os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main(sys.argv[1:])
