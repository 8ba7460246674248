"""Training driver shared by every GAN class (reference: vegans/models/unconditional/AbstractGenerativeModel.py
and vegans/models/conditional/AbstractConditionalGenerativeModel.py).

The public surface is the reference's: constructor arguments, `fit`, `get_training_results`, `get_losses`,
`generate`, `predict`, `sample`, `save`/`load`, `summary`, the `neural_nets / optimizers / steps / _losses /
logged_losses / hyperparameters` attributes and the subclass hooks `_default_optimizer`, `_define_loss`,
`calculate_losses`, `_zero_grad`, `_backward`, `_step`.  What changed is below that surface:

  * batches are cut from the numpy array directly (same order and partial last batch as the reference's
    unshuffled DataLoader), cast to float32 on the host and laid out NHWC on the device by a layout kernel;
  * every forward / backward / optimizer kernel is a HIP kernel of libvegans_hip.so (vegans.engine);
  * while one network is being trained the other networks' parameters are frozen for that sub-step, which skips
    weight gradients the reference computes and then throws away (their values never reach an optimizer);
  * with `ngpu > 1` under `torch.distributed` (one process per GPU) each rank trains on its contiguous shard of
    every batch and gradients are summed over RCCL before the fused optimizer step.
"""
import os
import sys
import time
from abc import ABC, abstractmethod
from collections import deque
from contextlib import contextmanager
from datetime import datetime

import numpy as np
import torch

from ..engine import graph, ops, optim as hip_optim, parallel, streams
from ..utils import utils


_COPY_STREAMS = {}      # device index -> side stream for batch uploads
_UPLOADS_IN_FLIGHT = 2  # iterations the host may be ahead of the GPU when it uploads another batch
_STAGING = {}           # device index -> {position of the array in a batch: _PinnedRing}


class _PinnedRing:
    """Page-locked staging buffers for the batches of `fit`, reused round robin. A fresh pinned allocation per batch
    (`Tensor.pin_memory()`) costs the host several milliseconds for a 12 MB batch whenever the allocator cannot hand a
    cached block back yet; here a batch is copied (and, for other numeric types, converted) straight into a buffer
    that is known to be free: the event recorded behind its previous upload has completed. Buffers are keyed by shape
    and type, so a ragged last batch gets its own."""

    SLOTS = _UPLOADS_IN_FLIGHT + 2
    MAX_SHAPES = 4          # batch shapes kept (a full batch, a ragged last one, a test batch ...): the oldest goes first

    def __init__(self):
        self.slots = {}      # (shape, dtype) -> [[tensor, numpy view, event or None], ...]
        self.turn = {}
        self.last = None

    def stage(self, array, dtype):
        key = (tuple(array.shape), np.dtype(dtype).str)
        if key not in self.slots and len(self.slots) >= self.MAX_SHAPES:
            oldest = next(iter(self.slots))
            for _, _, ev in self.slots.pop(oldest):      # its uploads have left the buffers before they are released
                if ev is not None:
                    ev.synchronize()
            self.turn.pop(oldest, None)
        ring = self.slots.setdefault(key, [])
        if len(ring) < self.SLOTS:
            t = torch.empty(tuple(array.shape), dtype=torch.float64 if np.dtype(dtype) == np.float64 else torch.float32,
                            pin_memory=True)
            ring.append([t, t.numpy(), None])
            slot = ring[-1]
        else:
            i = self.turn.get(key, 0)
            self.turn[key] = (i + 1) % self.SLOTS
            slot = ring[i]
            if slot[2] is not None:
                slot[2].synchronize()
        np.copyto(slot[1], array, casting="unsafe")
        self.last = slot
        return slot[0]

    def sent(self):
        """Call right after the asynchronous copy out of the buffer `stage` returned has been enqueued."""
        ev = torch.cuda.Event()
        ev.record()
        self.last[2] = ev


def _period(value, nr_batches):
    """'0.25e' -> every quarter epoch (at least every batch); ints pass through; None disables."""
    if isinstance(value, str):
        return max(int(float(value.split("e")[0]) * nr_batches), 1)
    return value


class AbstractGenerativeModel(ABC):
    # ------------------------------------------------------------------------------------------------------
    # construction
    # ------------------------------------------------------------------------------------------------------
    def __init__(self, x_dim, z_dim, optim, optim_kwargs, fixed_noise_size, device, folder, ngpu):
        self.x_dim = (x_dim,) if isinstance(x_dim, int) else tuple(x_dim)
        self.z_dim = (z_dim,) if isinstance(z_dim, int) else tuple(z_dim)
        self.ngpu = ngpu if ngpu is not None else 0
        self.fixed_noise_size = fixed_noise_size
        self.device = resolve_device(device)

        if not hasattr(self, "folder"):
            if folder is None:
                self.folder = ""
            else:
                # the reference keeps only the last character of a folder name ending in "/" (folder[-1]);
                # that quirk is preserved so existing scripts create the same directories
                folder = folder if not folder.endswith("/") else folder[-1]
                if os.path.exists(folder):
                    folder += datetime.now().strftime("%Y%m%d_%H%M%S")
                self.folder = folder + "/"
                os.makedirs(self.folder)

        self._define_loss()
        self._define_optimizers(optim=optim, optim_kwargs=optim_kwargs)
        self.to(self.device)

        self.images_produced = len(self.adversariat.input_size) > 1
        ops.unit_scalar(self.device)        # created outside any graph capture
        self.fixed_noise = self.sample(n=fixed_noise_size)
        self._check_attributes()
        self.hyperparameters = {
            "x_dim": x_dim, "z_dim": z_dim, "ngpu": ngpu, "folder": folder, "optimizers": self.optimizers,
            "device": self.device, "loss_functions": self.loss_functions,
        }
        self._init_run = True
        self.eval()

    def _define_optimizers(self, optim, optim_kwargs):
        self._opt_kwargs = {name: {} for name in self.neural_nets}
        if isinstance(optim_kwargs, dict):
            self._opt_kwargs.update(optim_kwargs)
        self._check_dict_keys(self._opt_kwargs, where="_define_optimizers_kwargs")

        if optim is None:
            self._opt = {name: self._default_optimizer() for name in self.neural_nets}
        elif isinstance(optim, dict):
            self._opt = dict(optim)
            for name in self.neural_nets:
                self._opt.setdefault(name, self._default_optimizer())
        else:
            self._opt = {name: optim for name in self.neural_nets}
        self._check_dict_keys(self._opt, where="_define_optimizers")

        self.optimizers = {
            name: self._opt[name](params=network.parameters(), **self._opt_kwargs[name])
            for name, network in self.neural_nets.items()
        }

    def _check_dict_keys(self, param_dict, where):
        for name in param_dict:
            if name not in self.neural_nets:
                raise KeyError("Error in {}: {} not in self.neural_nets. Must be one of: {}.".format(
                    where, name, list(self.neural_nets)))
        for name in self.neural_nets:
            if name not in param_dict:
                raise KeyError("Error in {}: self.{} not in param_dict.".format(where, name))

    def _check_attributes(self):
        assert hasattr(self, "neural_nets"), "Model must have attribute 'neural_nets'."
        assert hasattr(self, "device"), "Model must have attribute 'device'."
        assert hasattr(self, "optimizers"), "Model must have attribute 'optimizers'."
        assert isinstance(self.ngpu, int) and self.ngpu >= 0, "ngpu must be positive integer. Given: {}.".format(self.ngpu)
        assert len(self.z_dim) in (1, 3), (
            "z_dim must either have length 1 (for vector input) or 3 (for image input). Given: {}.".format(self.z_dim))
        assert len(self.x_dim) in (1, 3), (
            "x_dim must either have length 1 (for vector input) or 3 (for image input). Given: {}.".format(self.x_dim))
        assert isinstance(self.neural_nets, dict), "'neural_nets' attribute of AbstractGenerativeModel must be dictionary."
        self._check_dict_keys(self.optimizers, where="_define_optimizer_kwargs")

    @abstractmethod
    def _default_optimizer(self):
        pass

    @abstractmethod
    def _define_loss(self):
        pass

    # ------------------------------------------------------------------------------------------------------
    # training set-up
    # ------------------------------------------------------------------------------------------------------
    def _set_up_training(self, X_train, y_train, X_test, y_test, epochs, batch_size, steps, print_every,
                         save_model_every, save_images_every, save_losses_every, enable_tensorboard):
        self._assert_shapes(X_train=X_train, y_train=y_train, X_test=X_test, y_test=y_test)
        nr_test = 0 if X_test is None else len(X_test)
        nr_batches = utils.number_of_batches(len(X_train), batch_size)

        writer_train = writer_test = None
        if enable_tensorboard and parallel.is_main():
            writer_cls = _summary_writer_class()
            if writer_cls is not None:
                writer_train = writer_cls(self.folder + "tensorboard/train/")
                if X_test is not None:
                    writer_test = writer_cls(self.folder + "tensorboard/test/")

        self._create_steps(steps=steps)
        save_periods = self._set_up_saver(
            print_every=print_every, save_model_every=save_model_every, save_images_every=save_images_every,
            save_losses_every=save_losses_every, nr_batches=nr_batches)
        self.hyperparameters.update({
            "epochs": epochs, "batch_size": batch_size, "steps": self.steps, "print_every": print_every,
            "save_model_every": save_model_every, "save_images_every": save_images_every,
            "enable_tensorboard": enable_tensorboard, "nr_train": len(X_train), "nr_test": nr_test,
        })
        return nr_batches, writer_train, writer_test, save_periods

    def _assert_shapes(self, X_train, y_train, X_test, y_test):
        assert len(X_train.shape) in (2, 4), (
            "X_train must be either have 2 or 4 shape dimensions. Given: {}.".format(X_train.shape) +
            "Try to use X_train.reshape(-1, 1) or X_train.reshape(-1, 1, height, width).")
        assert X_train.shape[2:] == self.adversariat.input_size[1:], (
            "Wrong input shape for adversariat. Given: {}. Needed: {}.".format(X_train.shape, self.adversariat.input_size))
        if X_test is not None:
            assert X_train.shape[1:] == X_test.shape[1:], (
                "X_train and X_test must have same dimensions. Given: {} and {}.".format(X_train.shape[1:], X_test.shape[1:]))
            if y_test is not None:
                assert X_test.shape[0] == y_test.shape[0], (
                    "Same number if X_test and y_test needed.Given: {} and {}.".format(X_test.shape[0], y_test.shape[0]))
        if y_train is not None:
            assert X_train.shape[0] == y_train.shape[0], (
                "Same number if X_train and y_train needed.Given: {} and {}.".format(X_train.shape[0], y_train.shape[0]))
            assert len(y_train.shape) in (2, 4), (
                "y_train must be either have 2 or 4 shape dimensions. Given: {}.".format(y_train.shape) +
                "Try to use y_train.reshape(-1, 1) or y_train.reshape(-1, 1, height, width).")
            assert y_train.shape[2:] == self.y_dim[1:], (
                "Wrong input shape for y_train. Given: {}. Needed: {}.".format(y_train.shape, self.y_dim))
            if len(y_train.shape) == 4:
                assert X_train.shape == y_train.shape, (
                    "If y_train is an image (Image-to-Image translation task) it must have the same shape as X_train"
                    " to be concatenated before passing it to the discriminator. x_shape: {}. y_shape: {}.".format(
                        X_train.shape, y_train.shape))
            if X_test is not None:
                assert y_train.shape[1:] == y_test.shape[1:], (
                    "y_train and y_test must have same dimensions. Given: {} and {}.".format(y_train.shape[1:], y_test.shape[1:]))

    def _create_steps(self, steps):
        self.steps = {name: 1 for name in self.neural_nets}
        if steps is not None:
            assert isinstance(steps, dict), "steps parameter must be of type dict. Given: {}.".format(type(steps))
            self.steps = steps
            for name in self.neural_nets:
                self.steps.setdefault(name, 1)
            self._check_dict_keys(self.steps, where="_create_steps")

    def _set_up_saver(self, print_every, save_model_every, save_images_every, save_losses_every, nr_batches):
        print_every = _period(print_every, nr_batches)
        if save_model_every is not None:
            if parallel.is_main():
                os.mkdir(self.folder + "models/")
            save_model_every = _period(save_model_every, nr_batches)
        if save_images_every is not None:
            if parallel.is_main():
                os.mkdir(self.folder + "images/")
            save_images_every = _period(save_images_every, nr_batches)
        save_losses_every = _period(save_losses_every, nr_batches)
        self.total_training_time = 0
        self.current_timer = time.perf_counter()
        self.batch_training_times = []
        return print_every, save_model_every, save_images_every, save_losses_every

    # ------------------------------------------------------------------------------------------------------
    # host -> device
    # ------------------------------------------------------------------------------------------------------
    def _to_device(self, array, staging=None):
        """numpy batch (or tensor) -> float32 tensor on the GPU, NHWC in memory for images. float64 arrays (numpy's
        default) travel as they are and are rounded to float32 on the device, in the pass that re-lays images (what
        the reference's `.float()` does, AbstractGAN1v1.py:77). `staging`: a `_PinnedRing` whose buffers the batch is
        staged through instead of a fresh pinned allocation per call."""
        if isinstance(array, torch.Tensor):
            t = array
            if t.dtype != torch.float32:
                t = t.float()
            t = t.to(self.device)
            return ops.to_nhwc(t) if t.dim() == 4 else t
        # pinned staging + asynchronous copy: a copy from pageable memory blocks the host until the stream has
        # drained, i.e. until the previous iteration has finished, and leaves the GPU idle during the transfer
        array = np.asarray(array)
        keep = array.dtype if array.dtype in (np.float32, np.float64) else np.dtype(np.float32)
        if staging is not None:
            host = staging.stage(array, keep)
        else:
            host = torch.from_numpy(np.ascontiguousarray(array, dtype=keep)).pin_memory()
        t = host.to(self.device, non_blocking=True)
        if staging is not None:
            staging.sent()
        if t.dtype == torch.float64:
            return ops.from_float64(t)
        return ops.to_nhwc(t) if t.dim() == 4 else t

    def _device_batches(self, X_train, y_train, batch_size, dp):
        """The batches of one epoch as device tensors, in the reference's order (AbstractGenerativeModel.py:164-166,
        319), uploaded ahead of their use on a side stream: batch i+1 goes through pinned memory, the copy engine and
        the layout kernel while earlier iterations run. Yields X or (X, y).

        The side stream never waits for the training stream on the device. Its buffers come from its own pool of the
        caching allocator, which hands a block out again only after every stream recorded on it (`record_stream`
        below) has passed the point of release, so nothing the training stream still reads can be overwritten. A
        device-side wait here would make every upload start exactly when an iteration starts, and a queue that has just
        been released by another queue's signal dispatches its next few kernels tens of microseconds apart (measured:
        0.3 ms per iteration). How far the host may run ahead is bounded on the host instead: it blocks until the
        iteration `_UPLOADS_IN_FLIGHT` back has finished, which also bounds the device memory held by uploaded batches."""
        device = torch.cuda.current_device()
        copy_stream = _COPY_STREAMS.get(device)          # kept off the model object, which must stay picklable
        if copy_stream is None:
            copy_stream = _COPY_STREAMS[device] = torch.cuda.Stream()

        rings = _STAGING.setdefault(device, {})

        def upload(batch):
            parts = batch if isinstance(batch, tuple) else (batch,)
            with torch.cuda.stream(copy_stream):
                out = tuple(self._to_device(dp.shard(part), rings.setdefault(i, _PinnedRing()))
                            for i, part in enumerate(parts))
                done = torch.cuda.Event()
                done.record(copy_stream)
            return out, done, (dp.global_rows, getattr(dp, "local_rows", 0))

        def hand_over(item):
            out, done, rows = item
            # the sharding context describes the batch being trained on, not the one being uploaded
            dp.global_rows = rows[0]
            if hasattr(dp, "local_rows"):
                dp.local_rows = rows[1]
            main = torch.cuda.current_stream()
            main.wait_event(done)
            for t in out:
                t.record_stream(main)
            return out if len(out) > 1 else out[0]

        # once per epoch: whatever produced the training data on the training stream is finished before it is read here
        copy_stream.wait_stream(torch.cuda.current_stream())
        pending = None
        ends = deque()          # one event per iteration handed out, recorded once the caller has enqueued it
        for batch in utils.iterate_batches(X_train, y_train, batch_size):
            while len(ends) > _UPLOADS_IN_FLIGHT:
                ends.popleft().synchronize()
            nxt = upload(batch)
            if pending is not None:
                yield hand_over(pending)
                end = torch.cuda.Event()
                end.record(torch.cuda.current_stream())
                ends.append(end)
            pending = nxt
        if pending is not None:
            yield hand_over(pending)

    # ------------------------------------------------------------------------------------------------------
    # training
    # ------------------------------------------------------------------------------------------------------
    def fit(self, X_train, X_test=None, epochs=5, batch_size=32, steps=None, print_every="1e",
            save_model_every=None, save_images_every=None, save_losses_every="1e", enable_tensorboard=True):
        """Train the networks. Arguments as in the reference (AbstractGenerativeModel.py:265-297)."""
        if not getattr(self, "_init_run", False):
            raise ValueError("Run initializer of the AbstractGenerativeModel class is your subclass!")
        if parallel.needs_workers(self):
            # ngpu > 1 in a plain single process: one worker process per GPU trains, the result comes back into this
            # object (the reference fans out inside the process with nn.DataParallel, utils/networks.py:39-41)
            from ..engine import spawn
            return spawn.fit_in_workers(self, dict(
                X_train=X_train, X_test=X_test, epochs=epochs, batch_size=batch_size, steps=steps,
                print_every=print_every, save_model_every=save_model_every, save_images_every=save_images_every,
                save_losses_every=save_losses_every, enable_tensorboard=enable_tensorboard))
        nr_batches, writer_train, writer_test, save_periods = self._set_up_training(
            X_train, y_train=None, X_test=X_test, y_test=None, epochs=epochs, batch_size=batch_size, steps=steps,
            print_every=print_every, save_model_every=save_model_every, save_images_every=save_images_every,
            save_losses_every=save_losses_every, enable_tensorboard=enable_tensorboard)
        print_every, save_model_every, save_images_every, save_losses_every = save_periods
        test_x_batch = self._to_device(X_test[:batch_size]) if X_test is not None else None
        dp = parallel.context(self)

        self.train()
        if save_images_every is not None:
            self._log_images(images=self.generate(z=self.fixed_noise), step=0, writer=writer_train)
        for epoch in range(epochs):
            self._print("---" * 20)
            self._print("EPOCH:", epoch + 1)
            self._print("---" * 20)
            for batch, X in enumerate(self._device_batches(X_train, None, batch_size, dp)):
                batch += 1
                step = epoch * nr_batches + batch
                Z = self._draw_latent(len(X), dp)
                self.train_on_batch(X, Z)

                if print_every is not None and step % print_every == 0:
                    self._losses = {}
                    self.calculate_losses(X_batch=X, Z_batch=Z)
                    self._summarise_batch(batch=batch, max_batches=nr_batches, epoch=epoch, max_epochs=epochs,
                                          print_every=print_every)
                if save_model_every is not None and step % save_model_every == 0:
                    self.save(name="models/model_{}.torch".format(step))
                if save_images_every is not None and step % save_images_every == 0:
                    self._log_images(images=self.generate(z=self.fixed_noise), step=step, writer=writer_train)
                    self._save_losses_plot()
                if save_losses_every is not None and step % save_losses_every == 0:
                    self._log_losses(X_batch=X, Z_batch=Z, mode="Train")
                    if enable_tensorboard:
                        self._log_scalars(step=step, writer=writer_train)
                    if test_x_batch is not None:
                        self._log_losses(X_batch=test_x_batch, Z_batch=self.sample(n=len(test_x_batch)), mode="Test")
                        if enable_tensorboard:
                            self._log_scalars(step=step, writer=writer_test)

        self.eval()
        self._clean_up(writers=[writer_train, writer_test])

    def train_on_batch(self, X, Z):
        """The body of the hot loop (AbstractGenerativeModel.py:321-327) for one batch that already lives on the
        GPU: every network in `neural_nets` order, `steps[name]` sub-steps each, all on the same X and Z.  After a
        couple of eager calls the body is replayed from a captured HIP graph (engine/graph.py)."""
        graph.run_step(self, self._train_on_batch_body, (X, Z))

    def _train_on_batch_body(self, X, Z):
        ops.plane_cache.begin()
        ops.norm_bwd.sweep()
        try:
            for name in self.neural_nets:
                for _ in range(self.steps[name]):
                    self._losses = {}
                    with self._training_only(name):
                        self.calculate_losses(X_batch=X, Z_batch=Z, who=name)
                        self._zero_grad(who=name)
                        self._backward(who=name)
                    self._step(who=name)
        finally:
            ops.plane_cache.end()
            ops.norm_bwd.sweep()

    def _draw_latent(self, n_local, dp):
        """One latent batch per data batch (AbstractGenerativeModel.py:320). The training loop never needs
        d(loss)/dZ, so the tensor is detached. Under data parallelism each rank draws the whole batch from its own
        identically seeded generator and keeps its rows, so the union over ranks is the single-GPU batch."""
        if dp.world == 1:
            return self.sample(n=n_local).detach()
        return dp.shard(self.sample(n=dp.global_rows).detach())

    @contextmanager
    def _training_only(self, who):
        """Freeze every network except `who` for one sub-step: their weight gradients are not needed (the
        reference computes them and zeroes them before use, AbstractGenerativeModel.py:325)."""
        frozen = []
        if who is not None:
            for name, net in self.neural_nets.items():
                if name == who:
                    continue
                for p in net.parameters():
                    if p.requires_grad:
                        p.requires_grad_(False)
                        frozen.append(p)
        try:
            yield
        finally:
            for p in frozen:
                p.requires_grad_(True)

    @abstractmethod
    def calculate_losses(self, X_batch, Z_batch, who=None):
        pass

    def _zero_grad(self, who=None):
        if who is not None:
            self.optimizers[who].zero_grad()
        else:
            for optimizer in self.optimizers.values():
                optimizer.zero_grad()

    def _backward(self, who=None):
        """`loss.backward(retain_graph=True)` of AbstractGenerativeModel.py:367-372. Parameter gradients are collected
        by the engine's gradient ledger and added in a fixed order (engine/grads.py) instead of by autograd's
        accumulation; the seed gradient is a resident device scalar."""
        assert len(self._losses) != 0, "'self._losses' empty when performing '_backward'."
        seed = ops.unit_scalar(self.device)
        with parallel.open_ledger(self, who) as ledger:
            if who is not None:
                self._losses[who].backward(gradient=seed, retain_graph=True)
            else:
                for loss in self._losses.values():
                    if loss.requires_grad:
                        loss.backward(gradient=seed, retain_graph=True)
            # backward nodes ran on the streams of their forward passes (engine/streams.py); order the launch stream
            # behind all of them before the ledger adds up what they produced
            streams.join(after_backward=True)
            ledger.joined()

    def _step(self, who=None, clamp=None):
        names = [who] if who is not None else list(self.optimizers)
        for name in names:
            scale = parallel.reduce_gradients(self, name)
            hip_optim.fused_step(self.optimizers[name], grad_scale=scale, clamp=clamp if name == who else None)

    # ------------------------------------------------------------------------------------------------------
    # logging
    # ------------------------------------------------------------------------------------------------------
    def _print(self, *args):
        if parallel.is_main():
            print(*args)

    def _summarise_batch(self, batch, max_batches, epoch, max_epochs, print_every):
        step = epoch * max_batches + batch
        max_steps = max_epochs * max_batches
        remaining = max_steps - step
        self._print("Step: {} / {} (Epoch: {} / {}, Batch: {} / {})".format(
            step, max_steps, epoch + 1, max_epochs, batch, max_batches))
        self._print("---" * 20)
        for name, value in parallel.reduce_logged(self, self._losses).items():
            self._print("{}: {}".format(name, value))
        self.batch_training_times.append(time.perf_counter() - self.current_timer)
        self.total_training_time = np.sum(self.batch_training_times)
        time_per_batch = np.mean(self.batch_training_times) / print_every
        self._print("\n")
        self._print("Time left: ~{} minutes (Steps remaining: {}).".format(
            np.round(remaining * time_per_batch / 60, 3), remaining))
        self._print("\n")
        self.current_timer = time.perf_counter()

    def _log_images(self, images, step, writer):
        assert len(self.adversariat.input_size) > 1, (
            "Called _log_images in AbstractGenerativeModel for adversariat.input_size = {}.".format(
                self.adversariat.input_size))
        import matplotlib
        matplotlib.use("Agg", force=False)
        import matplotlib.pyplot as plt
        images = _as_numpy_nchw(images)
        if not parallel.is_main():
            return
        if writer is not None:
            writer.add_images("images", images[:, :3] if images.shape[1] >= 3 else images[:, :1], step)
        self._build_images(images)
        plt.savefig(self.folder + "images/image_{}.png".format(step))
        plt.close()
        self._print("Images logged.")

    @staticmethod
    def _build_images(images):
        import matplotlib.pyplot as plt
        images = _as_numpy_nchw(images)
        if images.ndim == 4:
            images = images.reshape((-1, *images.shape[-2:]))
        nrows = max(int(np.sqrt(len(images))), 1)
        ncols = max(len(images) // nrows, 1)
        fig, axs = plt.subplots(nrows=nrows, ncols=ncols, figsize=(10, 10))
        for ax, image in zip(np.ravel(axs), images):
            ax.imshow(image)
            ax.axis("off")
        return fig, axs

    def _log_losses(self, X_batch, Z_batch, mode):
        self.calculate_losses(X_batch=X_batch, Z_batch=Z_batch)
        self._append_losses(mode=mode)

    def _append_losses(self, mode):
        if not hasattr(self, "logged_losses"):
            self._create_logged_losses()
        # one device->host transfer per logging event (under data parallelism: one all-reduce of the row-weighted values)
        for name, value in parallel.reduce_logged(self, self._losses).items():
            self.logged_losses[mode][name].append(value)

    def _create_logged_losses(self):
        with_test = self.hyperparameters["nr_test"] is not None
        self.logged_losses = {"Train": {}}
        if with_test:
            self.logged_losses["Test"] = {}
        for name in self._losses:
            self.logged_losses["Train"][name] = []
            if with_test:
                self.logged_losses["Test"][name] = []

    def _save_losses_plot(self):
        if hasattr(self, "logged_losses") and parallel.is_main():
            import matplotlib.pyplot as plt
            utils.plot_losses(self.logged_losses, show=False, share=False)
            plt.savefig(self.folder + "losses.png")
            plt.close()

    def _log_scalars(self, step, writer):
        if writer is not None:
            for name, loss in self._losses.items():
                writer.add_scalar("Loss/{}".format(name), loss.item(), step)
            writer.add_scalar("Time/Total", self.total_training_time / 60, step)
            if len(self.batch_training_times):
                writer.add_scalar("Time/Batch", np.mean(self.batch_training_times) / 60, step)

    def _clean_up(self, writers=None):
        for writer in writers or []:
            if writer is not None:
                writer.close()

    # ------------------------------------------------------------------------------------------------------
    # results
    # ------------------------------------------------------------------------------------------------------
    def get_training_results(self, by_epoch=False, agg=None):
        """(samples generated from the fixed noise in eval mode, logged losses) — AbstractGenerativeModel.py:474-496."""
        samples = self.generate(self.fixed_noise)
        losses = self.get_losses(by_epoch=by_epoch, agg=agg)
        return samples, losses

    def get_losses(self, by_epoch=False, agg=None):
        if agg is None:
            agg = np.mean
        assert callable(agg), "agg: Aggregation function must be callable."
        losses_dict = self.logged_losses.copy()
        if by_epoch:
            epochs = self.get_hyperparameters()["epochs"]
            for mode, per_mode in losses_dict.items():
                for key, values in per_mode.items():
                    per_epoch = len(values) // epochs
                    per_mode[key] = [agg(values[e * per_epoch:(e + 1) * per_epoch]) for e in range(epochs)]
        return losses_dict

    def save(self, name=None):
        """Pickle the whole model object, as the reference does (AbstractGenerativeModel.py:534-546)."""
        if name is None:
            name = "model.torch"
        if parallel.is_main():      # under data parallelism the ranks hold identical models: one writes
            torch.save(self, self.folder + name)
        self._print("Model saved to {}.".format(self.folder + name))

    @staticmethod
    def load(path):
        return torch.load(path, weights_only=False)

    # ------------------------------------------------------------------------------------------------------
    # utilities
    # ------------------------------------------------------------------------------------------------------
    def sample(self, n):
        """Latent samples [n, *z_dim] from torch's generator of the device (AbstractGenerativeModel.py:568-581)."""
        return torch.randn(n, *self.z_dim, requires_grad=True, device=self.device)

    def generate(self, z=None, n=None):
        return self(z=z, n=n)

    def predict(self, x):
        if not isinstance(x, torch.Tensor):
            x = self._to_device(x)
        return self.adversariat(x)

    def get_hyperparameters(self):
        return self.hyperparameters

    def summary(self, save=False):
        out = open(self.folder + "summary.txt", "w") if save else None
        stdout = sys.stdout
        try:
            if out is not None:
                sys.stdout = out
            for name, net in self.neural_nets.items():
                print(net)
                net.summary()
                print("\n\n")
        finally:
            sys.stdout = stdout
            if out is not None:
                out.close()

    def eval(self):
        self._is_training = False
        for net in self.neural_nets.values():
            net.eval()

    def train(self):
        self._is_training = True
        for net in self.neural_nets.values():
            net.train()

    def to(self, device):
        for net in self.neural_nets.values():
            net.to(device)

    def __call__(self, z=None, n=None):
        if z is not None and n is not None:
            raise ValueError("Only one of 'z' and 'n' is needed.")
        if z is None and n is None:
            raise ValueError("Either 'z' or 'n' must be not None.")
        if n is not None:
            z = self.sample(n=n)
        sample = self.generator(z)
        if self._is_training:
            return sample
        return _as_numpy_nchw(sample)

    def __str__(self):
        self.summary()
        return type(self).__name__


# ----------------------------------------------------------------------------------------------------------
# conditional variant
# ----------------------------------------------------------------------------------------------------------
class AbstractConditionalGenerativeModel(AbstractGenerativeModel):
    def __init__(self, x_dim, z_dim, y_dim, optim, optim_kwargs, fixed_noise_size, device, folder, ngpu):
        utils.get_input_dim(dim1=x_dim, dim2=y_dim)
        utils.get_input_dim(dim1=z_dim, dim2=y_dim)
        AbstractGenerativeModel.__init__(
            self, x_dim=x_dim, z_dim=z_dim, optim=optim, optim_kwargs=optim_kwargs,
            fixed_noise_size=fixed_noise_size, device=device, folder=folder, ngpu=ngpu)
        # a list y_dim stays a list, as in the reference (AbstractConditionalGenerativeModel.py:54)
        self.y_dim = (y_dim,) if isinstance(y_dim, int) else y_dim
        self.hyperparameters["y_dim"] = self.y_dim
        self.eval()

    def _set_up_training(self, X_train, y_train, X_test, y_test, epochs, batch_size, steps, print_every,
                         save_model_every, save_images_every, save_losses_every, enable_tensorboard):
        out = super()._set_up_training(X_train, y_train, X_test, y_test, epochs, batch_size, steps, print_every,
                                       save_model_every, save_images_every, save_losses_every, enable_tensorboard)
        self.fixed_labels = self._to_device(y_train[:self.fixed_noise_size])
        return out

    def fit(self, X_train, y_train, X_test=None, y_test=None, epochs=5, batch_size=32, steps=None, print_every="1e",
            save_model_every=None, save_images_every=None, save_losses_every="1e", enable_tensorboard=True):
        """Train the conditional networks. Arguments as AbstractConditionalGenerativeModel.py:72-108."""
        if parallel.needs_workers(self):
            from ..engine import spawn
            return spawn.fit_in_workers(self, dict(
                X_train=X_train, y_train=y_train, X_test=X_test, y_test=y_test, epochs=epochs, batch_size=batch_size,
                steps=steps, print_every=print_every, save_model_every=save_model_every,
                save_images_every=save_images_every, save_losses_every=save_losses_every,
                enable_tensorboard=enable_tensorboard))
        nr_batches, writer_train, writer_test, save_periods = self._set_up_training(
            X_train, y_train, X_test=X_test, y_test=y_test, epochs=epochs, batch_size=batch_size, steps=steps,
            print_every=print_every, save_model_every=save_model_every, save_images_every=save_images_every,
            save_losses_every=save_losses_every, enable_tensorboard=enable_tensorboard)
        print_every, save_model_every, save_images_every, save_losses_every = save_periods
        test_x_batch = self._to_device(X_test[:batch_size]) if X_test is not None else None
        test_y_batch = self._to_device(y_test[:batch_size]) if X_test is not None else None
        dp = parallel.context(self)

        self.train()
        if save_images_every is not None:
            self._log_images(images=self.generate(y=self.fixed_labels, z=self.fixed_noise), step=0, writer=writer_train)
        for epoch in range(epochs):
            self._print("---" * 20)
            self._print("EPOCH:", epoch + 1)
            self._print("---" * 20)
            for batch, (X, y) in enumerate(self._device_batches(X_train, y_train, batch_size, dp)):
                batch += 1
                step = epoch * nr_batches + batch
                Z = self._draw_latent(len(X), dp)
                self.train_on_batch(X, y, Z)

                if print_every is not None and step % print_every == 0:
                    self._losses = {}
                    self.calculate_losses(X_batch=X, Z_batch=Z, y_batch=y)
                    self._summarise_batch(batch=batch, max_batches=nr_batches, epoch=epoch, max_epochs=epochs,
                                          print_every=print_every)
                if save_model_every is not None and step % save_model_every == 0:
                    self.save(name="models/model_{}.torch".format(step))
                if save_images_every is not None and step % save_images_every == 0:
                    self._log_images(images=self.generate(y=self.fixed_labels, z=self.fixed_noise), step=step,
                                     writer=writer_train)
                    self._save_losses_plot()
                if save_losses_every is not None and step % save_losses_every == 0:
                    # the reference logs the train images against the TEST labels here
                    # (AbstractConditionalGenerativeModel.py:162); kept, so a run without a test set fails the
                    # same way it does there
                    self._log_losses(X_batch=X, Z_batch=Z, y_batch=test_y_batch, mode="Train")
                    if enable_tensorboard:
                        self._log_scalars(step=step, writer=writer_train)
                    if test_x_batch is not None:
                        self._log_losses(X_batch=test_x_batch, Z_batch=self.sample(n=len(test_x_batch)),
                                         y_batch=test_y_batch, mode="Test")
                        if enable_tensorboard:
                            self._log_scalars(step=step, writer=writer_test)

        self.eval()
        self._clean_up(writers=[writer_train, writer_test])

    def train_on_batch(self, X, y, Z):
        """Conditional twin of the hot-loop body (AbstractConditionalGenerativeModel.py:135-141)."""
        graph.run_step(self, self._train_on_batch_body, (X, y, Z))

    def _train_on_batch_body(self, X, y, Z):
        ops.plane_cache.begin()
        ops.norm_bwd.sweep()
        try:
            for name in self.neural_nets:
                for _ in range(self.steps[name]):
                    self._losses = {}
                    with self._training_only(name):
                        self.calculate_losses(X_batch=X, Z_batch=Z, y_batch=y, who=name)
                        self._zero_grad(who=name)
                        self._backward(who=name)
                    self._step(who=name)
        finally:
            ops.plane_cache.end()
            ops.norm_bwd.sweep()

    def _log_losses(self, X_batch, Z_batch, y_batch, mode):
        self._losses = {}
        self.calculate_losses(X_batch=X_batch, Z_batch=Z_batch, y_batch=y_batch)
        self._append_losses(mode=mode)

    def get_training_results(self, by_epoch=False, agg=None):
        samples = self.generate(y=self.fixed_labels, z=self.fixed_noise)
        losses = self.get_losses(by_epoch=by_epoch, agg=agg)
        return samples, losses

    def concatenate(self, tensor1, tensor2):
        return utils.concatenate(tensor1=tensor1, tensor2=tensor2)

    def generate(self, y, z=None):
        return self(y=y, z=z)

    def predict(self, x, y):
        if not isinstance(x, torch.Tensor):
            x = self._to_device(x)
        if not isinstance(y, torch.Tensor):
            y = self._to_device(y)
        return self.adversariat(self.concatenate(x, y))

    def __call__(self, y, z=None):
        if z is None:
            z = self.sample(n=len(y))
        if not isinstance(y, torch.Tensor):
            y = self._to_device(y)
        sample = self.generator(self.concatenate(z, y))
        if self._is_training:
            return sample
        return _as_numpy_nchw(sample)


# ----------------------------------------------------------------------------------------------------------
def resolve_device(device):
    if device is None:
        device = "cuda" if torch.cuda.is_available() else "cpu"
    elif device not in ["cuda", "cpu"]:
        raise ValueError("device must be cuda or cpu.")
    if device == "cpu":
        raise RuntimeError("this build of vegans runs on MI355X only (device='cuda'); no GPU is visible or "
                           "device='cpu' was requested, and there is no CPU path.")
    return device


def _as_numpy_nchw(t):
    if isinstance(t, np.ndarray):
        return t
    t = t.detach()
    if t.dim() == 4:
        t = ops.to_nchw(t)
    return t.cpu().numpy()


def _summary_writer_class():
    try:
        from torch.utils.tensorboard import SummaryWriter
        return SummaryWriter
    except Exception:  # tensorboard is an optional logging dependency
        print("tensorboard is not installed: enable_tensorboard is ignored.")
        return None
