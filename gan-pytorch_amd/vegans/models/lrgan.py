"""Latent-regressor GANs: generator, discriminator and an encoder that recovers z from G(z)
(reference: vegans/models/unconditional/LRGAN.py, vegans/models/conditional/ConditionalLRGAN.py).

    generator   BCE(D(G(z)), 1) + lambda_L1 * L1(E(G(z)), z)
    adversary   0.5 * (BCE(D(G(z).detach()), 0) + BCE(D(x), 1))
    encoder     L1(E(G(z).detach()), z)
Training order inside a batch is generator, adversary, encoder.
"""
import torch

from ..engine import ops
from ..utils import utils
from ..utils.networks import Adversariat, Encoder, Generator
from .base import AbstractConditionalGenerativeModel, AbstractGenerativeModel, resolve_device


class _LatentRegressorLosses:
    def _default_optimizer(self):
        return torch.optim.Adam

    def _define_loss(self):
        self.loss_functions = {"Generator": utils.BCELoss(), "Adversariat": utils.BCELoss(), "L1": utils.L1Loss()}

    def encode(self, x):
        return self.encoder(x)

    def _dispatch(self, who, **batch):
        if who in (None, "Generator"):
            self._calculate_generator_loss(**batch)
        if who in (None, "Adversariat"):
            self._calculate_adversariat_loss(**batch)
        if who in (None, "Encoder"):
            self._calculate_encoder_loss(**batch)

    def _generator_terms(self, fake_images, fake_predictions, Z_batch):
        encoded_space = self.encode(x=fake_images)
        gen_loss_original = self.loss_functions["Generator"](fake_predictions, 1)
        latent_space_regression = self.loss_functions["L1"](encoded_space, Z_batch)
        gen_loss, _ = ops.combine(gen_loss_original, 1.0, latent_space_regression, float(self.lambda_L1))
        self._losses.update({
            "Generator": gen_loss,
            "Generator_Original": gen_loss_original,
            "Generator_L1": latent_space_regression,
        })

    def _adversariat_terms(self, fake_predictions, real_predictions):
        adv_loss_fake = self.loss_functions["Adversariat"](fake_predictions, 0)
        adv_loss_real = self.loss_functions["Adversariat"](real_predictions, 1)
        adv_loss, ratio = ops.combine(adv_loss_fake, 0.5, adv_loss_real, 0.5, want_ratio=True)
        self._losses.update({
            "Adversariat": adv_loss,
            "Adversariat_fake": adv_loss_fake,
            "Adversariat_real": adv_loss_real,
            "RealFakeRatio": ratio,
        })

    def _encoder_terms(self, fake_images, Z_batch):
        encoded_space = self.encoder(fake_images)
        self._losses.update({"Encoder": self.loss_functions["L1"](encoded_space, Z_batch)})

    def _check_shapes(self):
        assert self.generator.output_size == self.x_dim, (
            "Generator output shape must be equal to x_dim. {} vs. {}.".format(self.generator.output_size, self.x_dim))
        assert self.encoder.output_size == self.z_dim, (
            "Encoder output shape must be equal to z_dim. {} vs. {}.".format(self.encoder.output_size, self.z_dim))


class LRGAN(_LatentRegressorLosses, AbstractGenerativeModel):
    def __init__(self, generator, adversariat, encoder, x_dim, z_dim, optim=None, optim_kwargs=None, lambda_L1=10,
                 fixed_noise_size=32, device=None, folder="./AbstractGAN1v1", ngpu=0):
        device = resolve_device(device)
        self.generator = Generator(generator, input_size=z_dim, device=device, ngpu=ngpu)
        self.adversariat = Adversariat(adversariat, input_size=x_dim, adv_type="Discriminator", device=device, ngpu=ngpu)
        self.encoder = Encoder(encoder, input_size=x_dim, device=device, ngpu=ngpu)
        self.neural_nets = {"Generator": self.generator, "Adversariat": self.adversariat, "Encoder": self.encoder}
        AbstractGenerativeModel.__init__(
            self, x_dim=x_dim, z_dim=z_dim, optim=optim, optim_kwargs=optim_kwargs,
            fixed_noise_size=fixed_noise_size, device=device, folder=folder, ngpu=ngpu)
        self.lambda_L1 = lambda_L1
        self.hyperparameters["lambda_L1"] = lambda_L1
        self._check_shapes()

    def calculate_losses(self, X_batch, Z_batch, who=None):
        self._dispatch(who, X_batch=X_batch, Z_batch=Z_batch)

    def _calculate_generator_loss(self, X_batch, Z_batch):
        fake_images = self.generate(z=Z_batch)
        self._generator_terms(fake_images, self.predict(x=fake_images), Z_batch)

    def _calculate_adversariat_loss(self, X_batch, Z_batch):
        with torch.no_grad():
            fake_images = self.generate(z=Z_batch)
        self._adversariat_terms(self.predict(x=fake_images), self.predict(x=X_batch))

    def _calculate_encoder_loss(self, X_batch, Z_batch):
        with torch.no_grad():
            fake_images = self.generate(z=Z_batch)
        self._encoder_terms(fake_images, Z_batch)


class ConditionalLRGAN(_LatentRegressorLosses, AbstractConditionalGenerativeModel):
    def __init__(self, generator, adversariat, encoder, x_dim, z_dim, y_dim, optim=None, optim_kwargs=None,
                 lambda_L1=10, fixed_noise_size=32, device=None, folder="./AbstractGAN1v1", ngpu=0):
        adv_in_dim = utils.get_input_dim(dim1=x_dim, dim2=y_dim)
        gen_in_dim = utils.get_input_dim(dim1=z_dim, dim2=y_dim)
        device = resolve_device(device)
        self.generator = Generator(generator, input_size=gen_in_dim, device=device, ngpu=ngpu)
        self.adversariat = Adversariat(adversariat, input_size=adv_in_dim, adv_type="Discriminator", device=device,
                                       ngpu=ngpu)
        self.encoder = Encoder(encoder, input_size=x_dim, device=device, ngpu=ngpu)
        self.neural_nets = {"Generator": self.generator, "Adversariat": self.adversariat, "Encoder": self.encoder}
        AbstractConditionalGenerativeModel.__init__(
            self, x_dim=x_dim, z_dim=z_dim, y_dim=y_dim, optim=optim, optim_kwargs=optim_kwargs,
            fixed_noise_size=fixed_noise_size, device=device, folder=folder, ngpu=ngpu)
        self.lambda_L1 = lambda_L1
        self.hyperparameters["lambda_L1"] = lambda_L1
        self._check_shapes()

    def calculate_losses(self, X_batch, Z_batch, y_batch, who=None):
        self._dispatch(who, X_batch=X_batch, Z_batch=Z_batch, y_batch=y_batch)

    def _calculate_generator_loss(self, X_batch, Z_batch, y_batch):
        fake_images = self.generate(y=y_batch, z=Z_batch)
        self._generator_terms(fake_images, self.predict(x=fake_images, y=y_batch), Z_batch)

    def _calculate_adversariat_loss(self, X_batch, Z_batch, y_batch):
        with torch.no_grad():
            fake_images = self.generate(y=y_batch, z=Z_batch)
        self._adversariat_terms(self.predict(x=fake_images, y=y_batch), self.predict(x=X_batch, y=y_batch))

    def _calculate_encoder_loss(self, X_batch, Z_batch, y_batch):
        with torch.no_grad():
            fake_images = self.generate(y=y_batch, z=Z_batch)
        self._encoder_terms(fake_images, Z_batch)
