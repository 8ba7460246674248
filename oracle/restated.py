"""TEST INFRASTRUCTURE — CPU restatement of the VeGANs adversarial training step in plain torch (fp32, stock
CPU kernels).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; it is
the checker the HIP engine is compared against, never a fallback for it.

Parity status: PINNED.  `tests/golden/make_goldens.py` runs the reference itself (imported from
/root/reference in the build container) and this restatement on the same seeded inputs and requires bit-identical
losses, parameters, BatchNorm buffers and samples before it writes a fixture; `tests/test_oracle_golden.py`
re-checks the restatement against those committed fixtures wherever the tests run.

All arithmetic lives in the third-party dependency torch (the reference pins only `torch>=1.0.0`, setup.py:22;
fixtures were generated with torch 2.10.0).  Every function cites the reference lines it restates
(paths relative to the reference tree).
"""
import numpy as np
import torch
from torch import nn

ALGOS = ("vanilla", "wgan", "wgangp", "lsgan", "lrgan", "pix2pix")


# ---------------------------------------------------------------------------------------------------------------
# helpers restated from vegans/utils/utils.py
# ---------------------------------------------------------------------------------------------------------------
def wasserstein(pred, target_is_one):
    """vegans/utils/utils.py:45-49 — minus the mean for an all-ones target, plus the mean for all-zeros."""
    return -torch.mean(pred) if target_is_one else torch.mean(pred)


def concatenate(t1, t2):
    """vegans/utils/utils.py:52-90 — channel concat; a vector next to an image is tiled over the image plane,
    t1's channels first; dtype promotion is torch.cat's."""
    assert t1.shape[0] == t2.shape[0]
    n = t1.shape[0]
    if t1.dim() == 4 and t2.dim() == 2:
        t2 = t2.reshape(n, t2.shape[1], 1, 1).tile(1, 1, *t1.shape[2:])
    elif t1.dim() == 2 and t2.dim() == 4:
        t1 = t1.reshape(n, t1.shape[1], 1, 1).tile(1, 1, *t2.shape[2:])
    elif not (t1.dim() == t2.dim() and t1.dim() in (2, 4)):
        raise NotImplementedError("tensors must have 2 or 4 dimensions")
    return torch.cat((t1, t2), dim=1)


def input_dim(d1, d2):
    """vegans/utils/utils.py:92-128."""
    d1 = [d1] if isinstance(d1, int) else list(d1)
    d2 = [d2] if isinstance(d2, int) else list(d2)
    if len(d1) == 1 and len(d2) == 1:
        return [d1[0] + d2[0]]
    if len(d1) == 3 and len(d2) == 1:
        return [d1[0] + d2[0], d1[1], d1[2]]
    if len(d1) == 1 and len(d2) == 3:
        return [d2[0] + d1[0], d2[1], d2[2]]
    return [d1[0] + d2[0], d1[1], d1[2]]


def batches(X, y, batch_size):
    """vegans/utils/utils.py:9-20 + AbstractGenerativeModel.py:164-166 — what an unshuffled
    DataLoader(DataSet(X, y), batch_size) yields: consecutive rows, last partial batch kept, numpy -> tensor with
    the numpy dtype preserved.  Creating the DataLoader iterator draws one int64 "base seed" from torch's global
    generator (torch/utils/data/dataloader.py:706-708 of torch 2.10), once per epoch, before the first batch: that
    draw is restated here because it shifts every latent batch that follows."""
    torch.empty((), dtype=torch.int64).random_()
    for lo in range(0, len(X), batch_size):
        hi = min(lo + batch_size, len(X))
        xb = torch.from_numpy(np.stack([X[i] for i in range(lo, hi)]))
        if y is None:
            yield xb, None
        else:
            yield xb, torch.from_numpy(np.stack([y[i] for i in range(lo, hi)]))


def probe_output_shape(network, input_size):
    """vegans/utils/networks.py:92-94 — a batch of two torch.rand samples through the network in its current
    (train) mode; advances the torch RNG and the BatchNorm running statistics."""
    size = (input_size,) if isinstance(input_size, int) else tuple(input_size)
    return tuple(network(torch.rand([2, *size])).shape)


# ---------------------------------------------------------------------------------------------------------------
class OracleGAN:
    """One object restating constructor + fit() of the eleven reference classes.

    algo: vanilla | wgan | wgangp | lsgan | lrgan | pix2pix ; y_dim not None selects the Conditional* variant.
    Reference: AbstractGenerativeModel.py:23-89 (ctor), :265-354 (fit); AbstractGAN1v1.py:19-91;
    AbstractConditionalGAN1v1.py:21-87; AbstractConditionalGenerativeModel.py:16-66,72-174,254-282.
    """

    def __init__(self, algo, generator, adversariat, x_dim, z_dim, y_dim=None, encoder=None, optim=None,
                 optim_kwargs=None, fixed_noise_size=32, clip_val=0.01, lmbda_grad=10, lambda_L1=10):
        assert algo in ALGOS
        self.algo = algo
        self.conditional = y_dim is not None
        self.x_dim = (x_dim,) if isinstance(x_dim, int) else tuple(x_dim)
        self.z_dim = (z_dim,) if isinstance(z_dim, int) else tuple(z_dim)
        self.y_dim = y_dim
        self.clip_val, self.lmbda_grad, self.lambda_L1 = clip_val, lmbda_grad, lambda_L1
        gen_in = input_dim(z_dim, y_dim) if self.conditional else z_dim
        adv_in = input_dim(x_dim, y_dim) if self.conditional else x_dim
        # wrappers probe in construction order: generator, adversariat, encoder
        # (AbstractGAN1v1.py:35-36, LRGAN.py:50-52, networks.py:43)
        self.nets = {"Generator": generator, "Adversariat": adversariat}
        probe_output_shape(generator, gen_in)
        probe_output_shape(adversariat, adv_in)
        if algo == "lrgan":
            self.nets["Encoder"] = encoder
            probe_output_shape(encoder, x_dim)
        if self.conditional and algo != "lrgan":
            # the conditional 1v1 classes run the base constructor twice (AbstractConditionalGAN1v1.py:39-47):
            # the first pass builds optimizers and draws a fixed noise shaped like the CONCATENATED generator input,
            # the second pass rebuilds the optimizers and draws the fixed noise that is kept (raw z_dim)
            self._make_optimizers(optim, optim_kwargs)
            torch.randn(fixed_noise_size, *gen_in, requires_grad=True)
        self.optimizers = self._make_optimizers(optim, optim_kwargs)
        self.fixed_noise = self.sample(fixed_noise_size)
        self.training = False
        self._set_mode(False)

    # -- construction -------------------------------------------------------------------------------------
    def _default_optimizer(self):
        # VanillaGAN.py:52-53, LSGAN.py:49-50, LRGAN.py:70-71, ConditionalPix2Pix.py:54-55 -> Adam;
        # WassersteinGAN.py:61-62, WassersteinGANGP.py:61-62 -> RMSprop
        return torch.optim.RMSprop if self.algo in ("wgan", "wgangp") else torch.optim.Adam

    def _make_optimizers(self, optim, optim_kwargs):
        """AbstractGenerativeModel.py:91-117."""
        kwargs = {name: {} for name in self.nets}
        kwargs.update(optim_kwargs or {})
        if optim is None:
            classes = {name: self._default_optimizer() for name in self.nets}
        elif isinstance(optim, dict):
            classes = {name: optim.get(name, self._default_optimizer()) for name in self.nets}
        else:
            classes = {name: optim for name in self.nets}
        return {name: classes[name](params=net.parameters(), **kwargs[name]) for name, net in self.nets.items()}

    def sample(self, n):
        """AbstractGenerativeModel.py:568-581."""
        return torch.randn(n, *self.z_dim, requires_grad=True)

    def _set_mode(self, training):
        self.training = training
        for net in self.nets.values():
            net.train(training)

    # -- forward helpers ----------------------------------------------------------------------------------
    def generate(self, z, y=None):
        """AbstractGenerativeModel.py:664-674 / AbstractConditionalGenerativeModel.py:272-282 (train mode)."""
        if self.conditional:
            return self.nets["Generator"](concatenate(z, y).float())
        return self.nets["Generator"](z)

    def predict(self, x, y=None):
        """AbstractGenerativeModel.py:600-616 / AbstractConditionalGenerativeModel.py:254-270."""
        if self.conditional:
            return self.nets["Adversariat"](concatenate(x, y).float())
        return self.nets["Adversariat"](x)

    def _adv_loss_fn(self, pred, target_is_one):
        if self.algo in ("wgan", "wgangp"):
            return wasserstein(pred, target_is_one)
        target = torch.ones_like(pred) if target_is_one else torch.zeros_like(pred)
        if self.algo == "lsgan":
            return nn.functional.mse_loss(pred, target)          # LSGAN.py:52-53
        return nn.functional.binary_cross_entropy(pred, target)  # VanillaGAN.py:55-56 and BCE users

    # -- losses ---------------------------------------------------------------------------------------------
    def gradient_penalty(self, real, fake, alpha=None):
        """WassersteinGANGP.py:67-82 / ConditionalWassersteinGANGP.py:68-83. `alpha` (float64 numpy, shape
        [B,1,1,1]) may be injected; otherwise it is drawn from numpy's global generator like the reference."""
        if alpha is None:
            alpha = np.random.random((real.size(0), 1, 1, 1))
        alpha = torch.Tensor(alpha)
        interp = (alpha * real + ((1 - alpha) * fake)).requires_grad_(True).float()
        d_interp = self.nets["Adversariat"](interp)
        grads = torch.autograd.grad(outputs=d_interp, inputs=interp, grad_outputs=torch.ones_like(d_interp),
                                    create_graph=True, retain_graph=True, only_inputs=True)[0]
        grads = grads.view(grads.size(0), -1)
        return ((grads.norm(2, dim=1) - 1) ** 2).mean()

    def generator_loss(self, X, Z, y=None):
        """AbstractGAN1v1.py:66-72; LRGAN.py:95-111; ConditionalPix2Pix.py:60-74."""
        fake = self.generate(Z, y)
        pred = self.predict(fake, y)
        loss = self._adv_loss_fn(pred, True)
        out = {"Generator": loss}
        if self.algo == "lrgan":
            l1 = nn.functional.l1_loss(self.nets["Encoder"](fake), Z)
            out = {"Generator": loss + self.lambda_L1 * l1, "Generator_Original": loss, "Generator_L1": l1}
        elif self.algo == "pix2pix":
            l1 = nn.functional.l1_loss(X, fake)
            out = {"Generator": loss + 10 * l1, "Generator_Original": loss, "Generator_L1": l1}
        return out

    def adversariat_loss(self, X, Z, y=None, alpha=None):
        """AbstractGAN1v1.py:74-91; WassersteinGANGP.py:88-107; ConditionalWassersteinGANGP.py:89-111."""
        fake = self.generate(Z, y).detach()
        if self.algo == "wgangp" and not self.conditional:
            fake_pred = self.nets["Adversariat"](fake)
            real_pred = self.nets["Adversariat"](X.float())
        else:
            fake_pred = self.predict(fake, y)
            real_pred = self.predict(X if self.conditional else X.float(), y)
        l_fake = self._adv_loss_fn(fake_pred, False)
        l_real = self._adv_loss_fn(real_pred, True)
        out = {}
        if self.algo == "wgangp":
            if self.conditional:
                gp = self.gradient_penalty(concatenate(X, y), concatenate(fake, y), alpha)
            else:
                gp = self.gradient_penalty(X, fake, alpha)
            out["Adversariat"] = 0.5 * (l_fake + l_real) + self.lmbda_grad * gp
            out["Adversariat_grad"] = gp
        else:
            out["Adversariat"] = 0.5 * (l_fake + l_real)
        out["Adversariat_fake"] = l_fake
        out["Adversariat_real"] = l_real
        out["RealFakeRatio"] = l_real / l_fake
        return out

    def encoder_loss(self, X, Z, y=None):
        """LRGAN.py:132-140."""
        fake = self.generate(Z, y).detach()
        return {"Encoder": nn.functional.l1_loss(self.nets["Encoder"](fake), Z)}

    def losses(self, who, X, Z, y=None, alpha=None):
        if who == "Generator":
            return self.generator_loss(X, Z, y)
        if who == "Adversariat":
            return self.adversariat_loss(X, Z, y, alpha)
        return self.encoder_loss(X, Z, y)

    # -- one sub-step and the loop ---------------------------------------------------------------------------
    def sub_step(self, who, X, Z, y=None, alpha=None):
        """AbstractGenerativeModel.py:323-327 with _zero_grad/_backward/_step of :361-378 and the critic clamp of
        WassersteinGAN.py:74-81."""
        losses = self.losses(who, X, Z, y, alpha)
        self.optimizers[who].zero_grad()
        losses[who].backward(retain_graph=True)
        self.optimizers[who].step()
        if self.algo == "wgan" and who == "Adversariat":
            for p in self.nets["Adversariat"].parameters():
                p.data.clamp_(-self.clip_val, self.clip_val)
        return {k: float(v.detach()) for k, v in losses.items()}

    def sub_step_sharded(self, who, X, Z, y=None, alpha=None, shards=2):
        """The same sub-step as nn.DataParallel runs it over `shards` devices (utils/networks.py:39-41): the batch is
        cut into contiguous dim-0 chunks (torch.chunk), every chunk goes through the networks on its own, so BatchNorm
        uses per-chunk statistics; the loss is the mean over the whole batch, i.e. the chunk means weighted by
        rows/batch; gradients add up; only the first replica's BatchNorm buffers survive (the other replicas are
        thrown away after the forward pass). Emulated here on one module by running the chunks one after the other and
        restoring, at the end, the buffers as they were after the first chunk."""
        rows = len(X)
        bounds = []
        start = 0
        for piece in torch.chunk(torch.arange(rows), shards):
            bounds.append((start, start + len(piece)))
            start += len(piece)
        self.optimizers[who].zero_grad()
        totals, first_buffers = {}, None
        for lo, hi in bounds:
            part = self.losses(who, X[lo:hi], Z[lo:hi], None if y is None else y[lo:hi],
                               None if alpha is None else alpha[lo:hi])
            weight = (hi - lo) / float(rows)
            (part[who] * weight).backward(retain_graph=True)
            for k, v in part.items():
                totals[k] = totals.get(k, 0.0) + weight * float(v.detach())
            if first_buffers is None:
                first_buffers = {name: {k: b.detach().clone() for k, b in net.named_buffers()}
                                 for name, net in self.nets.items()}
        for name, net in self.nets.items():
            for k, b in net.named_buffers():
                b.data.copy_(first_buffers[name][k])
        self.optimizers[who].step()
        if self.algo == "wgan" and who == "Adversariat":
            for p in self.nets["Adversariat"].parameters():
                p.data.clamp_(-self.clip_val, self.clip_val)
        return totals

    def fit(self, X_train, y_train=None, epochs=1, batch_size=32, steps=None, latents=None, alphas=None, shards=1,
            X_test=None, y_test=None, print_every=None, save_losses_every=None):
        """The hot loop of AbstractGenerativeModel.py:312-327 / AbstractConditionalGenerativeModel.py:125-141.
        `latents` / `alphas` are optional iterators of injected Z batches / alpha arrays; when absent Z comes from
        torch.randn and alpha from numpy's global generator in the reference's order.
        `shards` > 1 runs every sub-step the way nn.DataParallel would over that many devices (sub_step_sharded).
        `print_every` / `save_losses_every` (ints, in batches) add the reference's logging passes
        (AbstractGenerativeModel.py:329-351, AbstractConditionalGenerativeModel.py:143-171): each is a full
        `calculate_losses(who=None)` in TRAIN mode, so it updates BatchNorm running statistics, draws one alpha for the
        gradient penalty and, for the test pass, one latent batch; the values land in `self.logged_losses`.
        Returns the list of per-sub-step loss dicts."""
        plan = {name: 1 for name in self.nets}
        plan.update(steps or {})
        latents = iter(latents) if latents is not None else None
        alphas = iter(alphas) if alphas is not None else None

        def draw_z(n):
            return next(latents) if latents is not None else self.sample(n)

        def draw_alpha():
            return next(alphas) if (alphas is not None and self.algo == "wgangp") else None

        test_x = test_y = None
        if X_test is not None:
            # `next(iter(test_dataloader))` (AbstractGenerativeModel.py:306; twice in
            # AbstractConditionalGenerativeModel.py:115-116): every DataLoader iterator draws a base seed from torch's
            # global generator before its first batch
            for _ in range(2 if self.conditional else 1):
                test_x, test_y = next(batches(X_test, y_test, batch_size))
        self.logged_losses = None
        history = []
        self._set_mode(True)
        step = 0
        for _ in range(epochs):
            for X, y in batches(X_train, y_train, batch_size):
                step += 1
                Z = draw_z(len(X))
                for name in self.nets:
                    for _ in range(plan[name]):
                        alpha = draw_alpha() if name == "Adversariat" else None
                        if shards > 1:
                            history.append((name, self.sub_step_sharded(name, X, Z, y, alpha, shards)))
                        else:
                            history.append((name, self.sub_step(name, X, Z, y, alpha)))
                if print_every is not None and step % print_every == 0:
                    self.all_losses(X, Z, y, draw_alpha(), shards)               # :329-335 (values printed only)
                if save_losses_every is not None and step % save_losses_every == 0:
                    # the conditional classes log the TRAIN images against the TEST labels
                    # (AbstractConditionalGenerativeModel.py:162)
                    self._log("Train", self.all_losses(X, Z, test_y if self.conditional else None, draw_alpha(), shards),
                              X_test is not None)
                    if test_x is not None:
                        self._log("Test", self.all_losses(test_x, draw_z(len(test_x)), test_y, draw_alpha(), shards), True)
        self._set_mode(False)
        return history

    def all_losses(self, X, Z, y=None, alpha=None, shards=1):
        """`calculate_losses(who=None)`: every network's loss on one batch, generator first
        (AbstractGAN1v1.py:57-64, LRGAN.py:83-93). `shards` > 1: as nn.DataParallel evaluates it, chunk by chunk with
        per-chunk BatchNorm statistics, the chunk means weighted by rows / batch (see sub_step_sharded); the buffers
        keep what the first chunk's pass left."""
        if shards <= 1:
            out = {}
            for who in self.nets:
                out.update(self.losses(who, X, Z, y, alpha if who == "Adversariat" else None))
            return {k: float(v.detach()) for k, v in out.items()}
        rows, start, bounds = len(X), 0, []
        for piece in torch.chunk(torch.arange(rows), shards):
            bounds.append((start, start + len(piece)))
            start += len(piece)
        totals, first_buffers = {}, None
        for lo, hi in bounds:
            part = {}
            for who in self.nets:
                part.update(self.losses(who, X[lo:hi], Z[lo:hi], None if y is None else y[lo:hi],
                                        alpha[lo:hi] if (alpha is not None and who == "Adversariat") else None))
            for k, v in part.items():
                totals[k] = totals.get(k, 0.0) + (hi - lo) / float(rows) * float(v.detach())
            if first_buffers is None:
                first_buffers = {name: {k: b.detach().clone() for k, b in net.named_buffers()}
                                 for name, net in self.nets.items()}
        for name, net in self.nets.items():
            for k, b in net.named_buffers():
                b.data.copy_(first_buffers[name][k])
        if "RealFakeRatio" in totals:
            totals["RealFakeRatio"] = totals["Adversariat_real"] / totals["Adversariat_fake"]
        return totals

    def _log(self, mode, values, with_test):
        """AbstractGenerativeModel.py:438-453. The "Test" key exists whenever logging happens: `nr_test` is 0, not None,
        without a test set (:158, :445)."""
        if self.logged_losses is None:
            self.logged_losses = {"Train": {k: [] for k in values}, "Test": {k: [] for k in values}}
        for k, v in values.items():
            self.logged_losses[mode][k].append(v)

    def fixed_samples(self, y=None):
        """AbstractGenerativeModel.py:474-496 — eval-mode samples from the fixed noise."""
        with torch.no_grad():
            return self.generate(self.fixed_noise, y).detach().cpu().numpy()
