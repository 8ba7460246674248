"""The CPU restatement (oracle/restated.py) against the golden vectors produced by the reference itself.
At generation time the two were bit-identical (tests/golden/make_goldens.py asserts it); here the check is
repeated wherever the suite runs, with a tolerance that allows for a different host CPU / thread count."""
import numpy as np
import pytest
import torch

import golden_util as gu
from oracle.restated import OracleGAN, concatenate, input_dim, wasserstein


@pytest.mark.parametrize("name", gu.names())
def test_oracle_reproduces_reference_fixture(name):
    fx = gu.load(name)
    case = fx["meta"]
    torch.set_num_threads(1)
    G, D, E = gu.build_nets(case)
    torch.manual_seed(case["seed"])
    np.random.seed(case["seed"])
    o = OracleGAN(case["algo"], G, D, case["x_dim"], case["z_dim"], y_dim=case.get("y_dim"), encoder=E,
                  optim_kwargs=case.get("optim_kwargs"), fixed_noise_size=case["fixed_noise_size"])
    # the constructor consumed the RNG exactly like the reference did
    np.testing.assert_array_equal(o.fixed_noise.detach().numpy(), fx["fixed_noise"])
    hist = o.fit(fx["X"], fx["y"], epochs=1, batch_size=case["batch"], steps=case.get("steps"),
                 latents=[torch.from_numpy(z) for z in fx["Z"]], alphas=list(fx["alpha"]), X_test=fx["X_test"],
                 y_test=fx["y_test"], **case.get("fit", {}))
    if fx["logged"] is not None:      # values of the logging passes (AbstractGenerativeModel.py:329-351)
        assert set(o.logged_losses) == set(fx["logged"])
        for mode, per_mode in fx["logged"].items():
            for k, want in per_mode.items():
                got = np.asarray(o.logged_losses[mode][k], dtype=np.float64)
                assert got.shape == want.shape, (mode, k)
                np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-6, equal_nan=True, err_msg="%s/%s" % (mode, k))
    rows = gu.loss_rows(fx)
    assert [w for w, _ in hist] == [w for w, _ in rows]
    for (who, got), (_, want) in zip(hist, rows):
        for k, v in want.items():
            if np.isnan(v):
                assert np.isnan(got[k])
            else:
                assert got[k] == pytest.approx(v, rel=1e-5, abs=1e-6), (who, k)
    for net, sd in fx["final"].items():
        state = o.nets[net].state_dict()
        for k, v in sd.items():
            np.testing.assert_allclose(state[k].numpy(), v, rtol=1e-5, atol=1e-6, err_msg="%s/%s" % (net, k))
    y = torch.from_numpy(fx["y"][:case["fixed_noise_size"]]) if fx["y"] is not None else None
    np.testing.assert_allclose(o.fixed_samples(y), fx["samples"], rtol=1e-5, atol=1e-6)


def test_wgan_clip_fixture_is_clipped_bit_exact():
    fx = gu.load("cwgan_clip_onehot")
    bound = np.float32(0.01)
    biggest = max(float(np.abs(v).max()) for k, v in fx["final"]["Adversariat"].items()
                  if not k.endswith(("running_mean", "running_var", "num_batches_tracked")))
    assert biggest == float(bound)


def test_concatenate_tiles_vector_over_image():
    x = torch.arange(2 * 1 * 2 * 2, dtype=torch.float32).reshape(2, 1, 2, 2)
    y = torch.tensor([[1.0, 0.0], [0.0, 1.0]])
    out = concatenate(x, y)
    assert out.shape == (2, 3, 2, 2)
    assert torch.equal(out[:, 0:1], x)
    assert torch.equal(out[0, 1], torch.ones(2, 2)) and torch.equal(out[0, 2], torch.zeros(2, 2))
    out = concatenate(y, x)           # vector first: its channels lead
    assert torch.equal(out[:, 2:3], x) and torch.equal(out[1, 1], torch.ones(2, 2))
    assert concatenate(y, y).shape == (2, 4)
    assert concatenate(y.double(), x).dtype == torch.float64   # dtype promotion as in torch.cat


def test_input_dim_and_wasserstein():
    assert input_dim([1, 4, 4], [10]) == [11, 4, 4]
    assert input_dim(8, 10) == [18]
    assert input_dim([4], [3, 8, 8]) == [7, 8, 8]
    assert input_dim([3, 8, 8], [3, 8, 8]) == [6, 8, 8]
    p = torch.tensor([[1.0], [3.0]])
    assert float(wasserstein(p, True)) == -2.0 and float(wasserstein(p, False)) == 2.0


def test_two_device_emulation_equals_single_device_without_batch_norm():
    """The oracle's nn.DataParallel emulation (sub_step_sharded) must be the plain sub-step when no layer couples the
    rows of a batch: chunk means weighted by rows/batch are the batch mean. (With BatchNorm it differs by design:
    per-chunk statistics.) Uneven chunks included."""
    import copy

    import numpy as np
    import torch
    from torch import nn

    import bench_configs as cfg
    from oracle.restated import OracleGAN

    def nets():
        G = cfg.seeded_init(cfg.dcgan_generator(nz=8, ngf=4, nc=3, size=16), 1, 0.1)
        G = nn.Sequential(*[m for m in G if not isinstance(m, nn.BatchNorm2d)])
        D = cfg.seeded_init(cfg.dcgan_critic(ndf=4, nc=3, size=16, batch_norm=False), 2, 0.1)
        return G, D

    X = cfg.synthetic_images(14, (3, 16, 16), seed=1)          # batches of 7: chunks of 4 and 3 rows
    Z = [torch.randn(7, 8, 1, 1, generator=torch.Generator().manual_seed(i)) for i in range(2)]
    A = [np.random.default_rng(i).random((7, 1, 1, 1)) for i in range(2)]
    finals = []
    for shards in (1, 2):
        torch.manual_seed(0)
        G, D = nets()
        o = OracleGAN("wgangp", G, D, (3, 16, 16), (8, 1, 1), fixed_noise_size=2)
        hist = o.fit(X, None, epochs=1, batch_size=7, latents=copy.deepcopy(Z), alphas=A, shards=shards)
        finals.append((hist, [p.detach().clone() for net in (G, D) for p in net.parameters()]))
    for (_, a), (_, b) in zip(finals[0][0], finals[1][0]):
        for k in a:
            if k != "RealFakeRatio":
                assert abs(a[k] - b[k]) <= 1e-5 * max(abs(a[k]), 1e-1), (k, a[k], b[k])
    for a, b in zip(finals[0][1], finals[1][1]):
        assert float((a - b).abs().max()) <= 2e-3 * max(float(a.abs().max()), 1e-3)
