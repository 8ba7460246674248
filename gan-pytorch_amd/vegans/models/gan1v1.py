"""One-generator / one-adversary GAN algorithms on the HIP engine.

Reference counterparts (all under vegans/models/): unconditional/AbstractGAN1v1.py, VanillaGAN.py,
WassersteinGAN.py, WassersteinGANGP.py, LSGAN.py and conditional/AbstractConditionalGAN1v1.py,
ConditionalVanillaGAN.py, ConditionalWassersteinGAN.py, ConditionalWassersteinGANGP.py, ConditionalLSGAN.py,
ConditionalPix2Pix.py.

Loss arithmetic is the reference's:
    generator   L(D(G(z)), 1)
    adversary   0.5 * (L(D(G(z).detach()), 0) + L(D(x), 1))   [+ lmbda_grad * gradient penalty]
with L = BCE (Vanilla), mean with sign (Wasserstein), least squares (LS).  Targets are passed to the loss
kernels as constants; no target tensors are materialised and nothing is read back to the host.
"""
import numpy as np
import torch

from ..engine import graph, ops, parallel, streams
from ..utils import utils
from ..utils.networks import Adversariat, Generator
from .base import AbstractConditionalGenerativeModel, AbstractGenerativeModel, resolve_device


class AbstractGAN1v1(AbstractGenerativeModel):
    def __init__(self, generator, adversariat, x_dim, z_dim, adv_type, optim=None, optim_kwargs=None,
                 fixed_noise_size=32, device=None, folder="./AbstractGAN1v1", ngpu=0):
        device = resolve_device(device)
        self.generator = Generator(generator, input_size=z_dim, device=device, ngpu=ngpu)
        self.adversariat = Adversariat(adversariat, input_size=x_dim, adv_type=adv_type, device=device, ngpu=ngpu)
        # dict order is the training order inside one batch: generator first
        self.neural_nets = {"Generator": self.generator, "Adversariat": self.adversariat}
        super().__init__(x_dim=x_dim, z_dim=z_dim, optim=optim, optim_kwargs=optim_kwargs,
                         fixed_noise_size=fixed_noise_size, device=device, folder=folder, ngpu=ngpu)
        if hasattr(self, "_is_conditional"):
            assert self.generator.output_size[1:] == self.x_dim[1:], (
                "Generator output shape must be equal to x_dim. {} vs. {}.".format(self.generator.output_size, self.x_dim))
        else:
            assert self.generator.output_size == self.x_dim, (
                "Generator output shape must be equal to x_dim. {} vs. {}.".format(self.generator.output_size, self.x_dim))

    # -- losses ---------------------------------------------------------------------------------------------
    def calculate_losses(self, X_batch, Z_batch, who=None):
        if who == "Generator":
            self._calculate_generator_loss(X_batch=X_batch, Z_batch=Z_batch)
        elif who == "Adversariat":
            self._calculate_adversariat_loss(X_batch=X_batch, Z_batch=Z_batch)
        else:
            self._calculate_generator_loss(X_batch=X_batch, Z_batch=Z_batch)
            self._calculate_adversariat_loss(X_batch=X_batch, Z_batch=Z_batch)

    def _fake_detached(self, Z_batch, **cond):
        """G(z) for the adversary step: values only (the reference records the graph and then detaches)."""
        with torch.no_grad():
            return self.generate(z=Z_batch, **cond)

    def _calculate_generator_loss(self, X_batch, Z_batch):
        fake_images = self.generate(z=Z_batch)
        fake_predictions = self.predict(x=fake_images)
        gen_loss = self.loss_functions["Generator"](fake_predictions, 1)
        self._losses.update({"Generator": gen_loss})

    def _publish_adversariat(self, adv_loss_fake, adv_loss_real, penalty=None, weight=0.0):
        # 0.5*(fake+real) [+ weight*penalty] and real/fake in one scalar kernel
        adv_loss, ratio = ops.combine(adv_loss_fake, 0.5, adv_loss_real, 0.5, penalty, weight, want_ratio=True)
        self._losses.update({
            "Adversariat": adv_loss,
            "Adversariat_fake": adv_loss_fake,
            "Adversariat_real": adv_loss_real,
        })
        if penalty is not None:
            self._losses["Adversariat_grad"] = penalty
        self._losses["RealFakeRatio"] = ratio

    def _calculate_adversariat_loss(self, X_batch, Z_batch):
        # the fake and the real pass share nothing but read-only weights: the real pass runs on a stream of its own
        fake_images = self._fake_detached(Z_batch)
        with streams.region() as fork:
            adv_loss_fake = self.loss_functions["Adversariat"](self.predict(x=fake_images), 0)
            with fork.branch(0):
                adv_loss_real = self.loss_functions["Adversariat"](self.predict(x=X_batch), 1)
        self._publish_adversariat(adv_loss_fake, adv_loss_real)


# -- loss selectors ---------------------------------------------------------------------------------------------
class _VanillaMixin:
    def _default_optimizer(self):
        return torch.optim.Adam

    def _define_loss(self):
        self.loss_functions = {"Generator": utils.BCELoss(), "Adversariat": utils.BCELoss()}


class _LeastSquaresMixin:
    def _default_optimizer(self):
        return torch.optim.Adam

    def _define_loss(self):
        self.loss_functions = {"Generator": utils.MSELoss(), "Adversariat": utils.MSELoss()}


class _WassersteinMixin:
    def _default_optimizer(self):
        return torch.optim.RMSprop

    def _define_loss(self):
        self.loss_functions = {"Generator": utils.wasserstein_loss, "Adversariat": utils.wasserstein_loss}


class _ClipMixin:
    """After every critic update clamp all critic parameters to [-clip_val, clip_val] (WassersteinGAN.py:74-81);
    the clamp runs inside the optimizer kernel.  Not applied when every network steps at once (who=None)."""

    def _step(self, who=None):
        if who == "Adversariat":
            clip = float(self._clip_val)
            AbstractGenerativeModel._step(self, who=who, clamp=(-clip, clip))
        else:
            AbstractGenerativeModel._step(self, who=who)


class _PenaltyMixin:
    """Gradient penalty of WassersteinGANGP.py:67-82: one uniform alpha per sample from numpy's global generator,
    interpolate real and fake, differentiate the critic output w.r.t. the interpolates (keeping the graph), and
    penalise (||grad||_2 - 1)^2 averaged over the batch."""

    def _define_loss(self):
        self.loss_functions = {"Generator": utils.wasserstein_loss, "Adversariat": utils.wasserstein_loss,
                               "GP": self._gradient_penalty}

    def _draw_alpha(self, n_local):
        dp = getattr(self, "_dp", None)
        if dp is not None and dp.world > 1:
            # every rank draws the alphas of the whole batch (same numpy seed) and keeps its rows
            def draw():
                return dp.shard(np.random.random((dp.global_rows, 1, 1, 1)))
        else:
            def draw():
                return np.random.random((n_local, 1, 1, 1))
        return graph.host_input((n_local, 1, 1, 1), draw, self.device)

    def _gradient_penalty(self, real_images, fake_images):
        rows = real_images.size(0)
        alpha = self._draw_alpha(rows)
        if real_images.dim() == 2:
            # vector data: the reference's [B,1,1,1] alpha broadcasts against [B,D] to all B x B (alpha, sample)
            # pairs, and its norm runs over the B*D gradient elements that belong to one alpha
            # (WassersteinGANGP.py:68-69,80-81). Under data parallelism a rank holds the alphas of its rows and pairs
            # them with the samples of the WHOLE batch, as nn.DataParallel does when it scatters the [B, 1, B, D]
            # critic input along its first axis: the other ranks' rows are gathered (no gradient flows into them)
            real_images = parallel.gather_rows(self, real_images)
            fake_images = parallel.gather_rows(self, fake_images)
            interpolates = ops.gp_interpolate_pairs(alpha, real_images, fake_images)
        else:
            interpolates = ops.gp_interpolate(alpha, real_images, fake_images)
        d_interpolates = self.adversariat(interpolates)
        ones = torch.empty_like(d_interpolates)
        ops.fill_(ones, 1.0)
        with ops.inputs_only():
            gradients = torch.autograd.grad(outputs=d_interpolates, inputs=interpolates, grad_outputs=ones,
                                            create_graph=True, retain_graph=True, only_inputs=True)[0]
        return ops.gradient_penalty(gradients, rows=rows)


# -- unconditional algorithms -------------------------------------------------------------------------------------
class VanillaGAN(_VanillaMixin, AbstractGAN1v1):
    def __init__(self, generator, adversariat, x_dim, z_dim, optim=None, optim_kwargs=None, fixed_noise_size=32,
                 device=None, folder="./VanillaGAN", ngpu=None):
        super().__init__(generator=generator, adversariat=adversariat, x_dim=x_dim, z_dim=z_dim,
                         adv_type="Discriminator", optim=optim, optim_kwargs=optim_kwargs,
                         fixed_noise_size=fixed_noise_size, device=device, folder=folder, ngpu=ngpu)


class LSGAN(_LeastSquaresMixin, AbstractGAN1v1):
    def __init__(self, generator, adversariat, x_dim, z_dim, optim=None, optim_kwargs=None, fixed_noise_size=32,
                 device=None, folder="./LSGAN", ngpu=None):
        super().__init__(generator=generator, adversariat=adversariat, x_dim=x_dim, z_dim=z_dim,
                         adv_type="Discriminator", optim=optim, optim_kwargs=optim_kwargs,
                         fixed_noise_size=fixed_noise_size, device=device, folder=folder, ngpu=ngpu)


class WassersteinGAN(_ClipMixin, _WassersteinMixin, AbstractGAN1v1):
    def __init__(self, generator, adversariat, x_dim, z_dim, optim=None, optim_kwargs=None, fixed_noise_size=32,
                 clip_val=0.01, device=None, folder="./WassersteinGAN", ngpu=None):
        super().__init__(generator=generator, adversariat=adversariat, x_dim=x_dim, z_dim=z_dim, adv_type="Critic",
                         optim=optim, optim_kwargs=optim_kwargs, fixed_noise_size=fixed_noise_size, device=device,
                         folder=folder, ngpu=ngpu)
        self._clip_val = clip_val
        self.hyperparameters["clip_val"] = clip_val


class WassersteinGANGP(_PenaltyMixin, _WassersteinMixin, AbstractGAN1v1):
    def __init__(self, generator, adversariat, x_dim, z_dim, optim=None, optim_kwargs=None, fixed_noise_size=32,
                 lmbda_grad=10, device=None, folder="./WassersteinGANGP", ngpu=None):
        super().__init__(generator=generator, adversariat=adversariat, x_dim=x_dim, z_dim=z_dim, adv_type="Critic",
                         optim=optim, optim_kwargs=optim_kwargs, fixed_noise_size=fixed_noise_size, device=device,
                         folder=folder, ngpu=ngpu)
        self.lmbda_grad = lmbda_grad
        self.hyperparameters["lmbda_grad"] = lmbda_grad

    def _calculate_adversariat_loss(self, X_batch, Z_batch):
        # three independent critic passes (fake, real, interpolates with their double backward): the launch stream and
        # two side streams
        fake_images = self._fake_detached(Z_batch)
        with streams.region() as fork:
            adv_loss_fake = self.loss_functions["Adversariat"](self.adversariat(fake_images), 0)
            with fork.branch(0):
                adv_loss_real = self.loss_functions["Adversariat"](self.adversariat(X_batch), 1)
            with fork.branch(1):
                adv_loss_grad = self.loss_functions["GP"](X_batch, fake_images)
        self._publish_adversariat(adv_loss_fake, adv_loss_real, penalty=adv_loss_grad, weight=self.lmbda_grad)


# -- conditional algorithms -----------------------------------------------------------------------------------------
class AbstractConditionalGAN1v1(AbstractConditionalGenerativeModel, AbstractGAN1v1):
    def __init__(self, generator, adversariat, x_dim, z_dim, y_dim, adv_type, optim=None, optim_kwargs=None,
                 fixed_noise_size=32, device=None, folder="./AbstractGAN1v1", ngpu=0):
        adv_in_dim = utils.get_input_dim(dim1=x_dim, dim2=y_dim)
        gen_in_dim = utils.get_input_dim(dim1=z_dim, dim2=y_dim)
        self._is_conditional = True
        # First pass wires the networks against the concatenated input widths, second pass restores the raw
        # x/z dims and draws the fixed noise with the raw z_dim (AbstractConditionalGAN1v1.py:39-47): optimizers
        # and fixed noise are therefore created twice, in the same order as the reference.
        AbstractGAN1v1.__init__(
            self, generator=generator, adversariat=adversariat, x_dim=adv_in_dim, z_dim=gen_in_dim,
            adv_type=adv_type, optim=optim, optim_kwargs=optim_kwargs, fixed_noise_size=fixed_noise_size,
            device=device, folder=folder, ngpu=0)
        AbstractConditionalGenerativeModel.__init__(
            self, x_dim=x_dim, z_dim=z_dim, y_dim=y_dim, optim=optim, optim_kwargs=optim_kwargs,
            fixed_noise_size=fixed_noise_size, device=device, folder=folder, ngpu=ngpu)

    def calculate_losses(self, X_batch, Z_batch, y_batch, who=None):
        if who == "Generator":
            self._calculate_generator_loss(X_batch=X_batch, Z_batch=Z_batch, y_batch=y_batch)
        elif who == "Adversariat":
            self._calculate_adversariat_loss(X_batch=X_batch, Z_batch=Z_batch, y_batch=y_batch)
        else:
            self._calculate_generator_loss(X_batch=X_batch, Z_batch=Z_batch, y_batch=y_batch)
            self._calculate_adversariat_loss(X_batch=X_batch, Z_batch=Z_batch, y_batch=y_batch)

    def _calculate_generator_loss(self, X_batch, Z_batch, y_batch):
        fake_images = self.generate(y=y_batch, z=Z_batch)
        fake_predictions = self.predict(x=fake_images, y=y_batch)
        gen_loss = self.loss_functions["Generator"](fake_predictions, 1)
        self._losses.update({"Generator": gen_loss})

    def _calculate_adversariat_loss(self, X_batch, Z_batch, y_batch):
        fake_images = self._fake_detached(Z_batch, y=y_batch)
        with streams.region() as fork:
            adv_loss_fake = self.loss_functions["Adversariat"](self.predict(x=fake_images, y=y_batch), 0)
            with fork.branch(0):
                adv_loss_real = self.loss_functions["Adversariat"](self.predict(x=X_batch, y=y_batch), 1)
        self._publish_adversariat(adv_loss_fake, adv_loss_real)


class ConditionalVanillaGAN(_VanillaMixin, AbstractConditionalGAN1v1):
    def __init__(self, generator, adversariat, x_dim, z_dim, y_dim, optim=None, optim_kwargs=None,
                 fixed_noise_size=32, device=None, folder="./ConditionalVanillaGAN", ngpu=None):
        super().__init__(generator=generator, adversariat=adversariat, x_dim=x_dim, z_dim=z_dim, y_dim=y_dim,
                         adv_type="Discriminator", optim=optim, optim_kwargs=optim_kwargs,
                         fixed_noise_size=fixed_noise_size, device=device, folder=folder, ngpu=ngpu)


class ConditionalLSGAN(_LeastSquaresMixin, AbstractConditionalGAN1v1):
    def __init__(self, generator, adversariat, x_dim, z_dim, y_dim, optim=None, optim_kwargs=None,
                 fixed_noise_size=32, device=None, folder="./ConditionalLSGAN", ngpu=None):
        super().__init__(generator=generator, adversariat=adversariat, x_dim=x_dim, z_dim=z_dim, y_dim=y_dim,
                         adv_type="Discriminator", optim=optim, optim_kwargs=optim_kwargs,
                         fixed_noise_size=fixed_noise_size, device=device, folder=folder, ngpu=ngpu)


class ConditionalWassersteinGAN(_ClipMixin, _WassersteinMixin, AbstractConditionalGAN1v1):
    def __init__(self, generator, adversariat, x_dim, z_dim, y_dim, optim=None, optim_kwargs=None,
                 fixed_noise_size=32, clip_val=0.01, device=None, folder="./ConditionalWassersteinGAN", ngpu=None):
        super().__init__(generator=generator, adversariat=adversariat, x_dim=x_dim, z_dim=z_dim, y_dim=y_dim,
                         adv_type="Critic", optim=optim, optim_kwargs=optim_kwargs,
                         fixed_noise_size=fixed_noise_size, device=device, folder=folder, ngpu=ngpu)
        self._clip_val = clip_val
        self.hyperparameters["clip_val"] = clip_val


class ConditionalWassersteinGANGP(_PenaltyMixin, _WassersteinMixin, AbstractConditionalGAN1v1):
    def __init__(self, generator, adversariat, x_dim, z_dim, y_dim, optim=None, optim_kwargs=None,
                 fixed_noise_size=32, lmbda_grad=10, device=None, folder="./ConditionalWassersteinGANGP", ngpu=None):
        super().__init__(generator=generator, adversariat=adversariat, x_dim=x_dim, z_dim=z_dim, y_dim=y_dim,
                         adv_type="Critic", optim=optim, optim_kwargs=optim_kwargs,
                         fixed_noise_size=fixed_noise_size, device=device, folder=folder, ngpu=ngpu)
        self.lmbda_grad = lmbda_grad
        self.hyperparameters["lmbda_grad"] = lmbda_grad

    def _calculate_adversariat_loss(self, X_batch, Z_batch, y_batch):
        fake_images = self._fake_detached(Z_batch, y=y_batch)
        fake_concat = self.concatenate(fake_images, y_batch)
        real_concat = self.concatenate(X_batch, y_batch)
        with streams.region() as fork:
            adv_loss_fake = self.loss_functions["Adversariat"](self.adversariat(fake_concat), 0)
            with fork.branch(0):
                adv_loss_real = self.loss_functions["Adversariat"](self.adversariat(real_concat), 1)
            with fork.branch(1):
                # the penalty interpolates the label-augmented inputs (ConditionalWassersteinGANGP.py:100-103)
                adv_loss_grad = self.loss_functions["GP"](real_concat, fake_concat)
        self._publish_adversariat(adv_loss_fake, adv_loss_real, penalty=adv_loss_grad, weight=self.lmbda_grad)


class ConditionalPix2Pix(_VanillaMixin, AbstractConditionalGAN1v1):
    """BCE adversarial loss plus an L1 term between the real image and G(z, y). The weight of the L1 term is 10
    whatever `lambda_l1` says, as in the reference (ConditionalPix2Pix.py:38,51)."""

    def __init__(self, generator, adversariat, x_dim, z_dim, y_dim, optim=None, optim_kwargs=None, lambda_l1=10,
                 fixed_noise_size=32, device=None, folder="./ConditionalPix2Pix", ngpu=None):
        super().__init__(generator=generator, adversariat=adversariat, x_dim=x_dim, z_dim=z_dim, y_dim=y_dim,
                         adv_type="Discriminator", optim=optim, optim_kwargs=optim_kwargs,
                         fixed_noise_size=fixed_noise_size, device=device, folder=folder, ngpu=ngpu)
        self.lambda_l1 = 10
        self.hyperparameters["lambda_l1"] = self.lambda_l1

    def _define_loss(self):
        self.loss_functions = {"Generator": utils.BCELoss(), "Adversariat": utils.BCELoss(), "L1": utils.L1Loss()}

    def _calculate_generator_loss(self, X_batch, Z_batch, y_batch):
        fake_images = self.generate(y=y_batch, z=Z_batch)
        fake_predictions = self.predict(x=fake_images, y=y_batch)
        gen_loss_original = self.loss_functions["Generator"](fake_predictions, 1)
        gen_loss_pixel_wise = self.loss_functions["L1"](X_batch, fake_images)
        gen_loss, _ = ops.combine(gen_loss_original, 1.0, gen_loss_pixel_wise, float(self.lambda_l1))
        self._losses.update({
            "Generator": gen_loss,
            "Generator_Original": gen_loss_original,
            "Generator_L1": gen_loss_pixel_wise,
        })
