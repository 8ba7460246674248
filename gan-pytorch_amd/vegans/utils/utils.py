"""User-facing helpers of the VeGANs API (reference: vegans/utils/utils.py).

`wasserstein_loss`, `concatenate` and `get_input_dim` are on the training path and run on the HIP engine;
`DataSet` keeps the reference's indexing contract.  Plot helpers are thin matplotlib conveniences and are not
part of the accelerated path.
"""
import numpy as np
import torch

from ..engine import ops


class DataSet(torch.utils.data.Dataset):
    """Indexable (X[, y]) pair, as vegans/utils/utils.py:9-20."""

    def __init__(self, X, y=None):
        self.X = X
        self.y = y

    def __len__(self):
        return len(self.X)

    def __getitem__(self, index):
        if self.y is None:
            return self.X[index]
        return self.X[index], self.y[index]


def iterate_batches(X, y, batch_size):
    """Sequential, unshuffled batches with the last partial batch kept: the order a default
    `DataLoader(DataSet(X, y), batch_size)` yields (AbstractGenerativeModel.py:164-166), without the per-sample
    Python collate."""
    n = len(X)
    for start in range(0, n, batch_size):
        stop = min(start + batch_size, n)
        if y is None:
            yield X[start:stop]
        else:
            yield X[start:stop], y[start:stop]


def number_of_batches(n, batch_size):
    return (n + batch_size - 1) // batch_size


def wasserstein_loss(input, target):
    """-mean(input) for an all-ones target, +mean(input) for an all-zeros target, otherwise None
    (vegans/utils/utils.py:45-49).  `target` may also be the number 1 or 0, which avoids the device->host
    round trip the tensor form needs to inspect its contents; the training loop uses the number form."""
    if isinstance(target, torch.Tensor):
        flags = torch.stack([(target == 1).all(), (target == 0).all()]).tolist()
        if flags[0]:
            return ops.loss_mean(input, -1.0)
        if flags[1]:
            return ops.loss_mean(input, 1.0)
        return None
    if target == 1:
        return ops.loss_mean(input, -1.0)
    if target == 0:
        return ops.loss_mean(input, 1.0)
    return None


class BCELoss:
    """mean binary cross entropy with torch's -100 log clamp; target: tensor or number."""

    def __call__(self, input, target):
        return ops.loss_bce(input, target)

    def __repr__(self):
        return "BCELoss()"


class MSELoss:
    def __call__(self, input, target):
        return ops.loss_mse(input, target)

    def __repr__(self):
        return "MSELoss()"


class L1Loss:
    def __call__(self, input, target):
        return ops.loss_l1(input, target)

    def __repr__(self):
        return "L1Loss()"


def concatenate(tensor1, tensor2):
    """Concatenate along the channel / feature axis; a 2-d tensor next to a 4-d one is tiled over the image
    plane first (vegans/utils/utils.py:52-90).  tensor1's channels come first."""
    if tensor1.shape[0] != tensor2.shape[0]:
        raise AssertionError("Tensors to concatenate must have same dim 0. Tensor1: {}. Tensor2: {}.".format(
            tensor1.shape[0], tensor2.shape[0]))
    if tensor1.dim() not in (2, 4) or tensor2.dim() not in (2, 4):
        raise NotImplementedError("tensor1 and tensor2 must have 2 or 4 dimensions. Given: {} and {}.".format(
            tensor1.shape, tensor2.shape))
    return ops.concatenate(_f32(tensor1), _f32(tensor2))


def _f32(t):
    # the reference concatenates in the promoted dtype and casts with .float() afterwards
    # (AbstractConditionalGenerativeModel.py:269,278); casting first gives the same float32 values
    return t if t.dtype == torch.float32 else t.float()


def get_input_dim(dim1, dim2):
    """Shape (without batch axis) of `concatenate` applied to inputs of shapes dim1 and dim2
    (vegans/utils/utils.py:92-128).  Ints are accepted for either argument (the reference only copes with an int
    dim1: its dim2 check at :116 tests the already converted dim1)."""
    d1 = [dim1] if isinstance(dim1, int) else list(dim1)
    d2 = [dim2] if isinstance(dim2, int) else list(dim2)
    if len(d1) == 1 and len(d2) == 1:
        return [d1[0] + d2[0]]
    if len(d1) == 3 and len(d2) == 1:
        return [d1[0] + d2[0], d1[1], d1[2]]
    if len(d1) == 1 and len(d2) == 3:
        return [d2[0] + d1[0], d2[1], d2[2]]
    assert d1[1] == d2[1] and d1[2] == d2[2], (
        "If both dim1 and dim2 are arrays, must have same shape. dim1: {}. dim2: {}.".format(dim1, dim2))
    return [d1[0] + d2[0], d1[1], d1[2]]


def plot_losses(losses, show=True, share=False):
    """Line plots of logged losses (matplotlib; convenience only)."""
    import matplotlib.pyplot as plt
    if share:
        fig, ax = plt.subplots(1, 1, figsize=(8, 8))
        for mode, per_mode in losses.items():
            for name, values in per_mode.items():
                ax.plot(values, lw=2, label=mode + name)
        ax.set_xlabel("Iterations")
        ax.legend()
        axs = ax
    else:
        n = len(losses["Train"])
        nrows = max(int(np.sqrt(n)), 1)
        ncols = max(n // nrows, 1)
        fig, axs = plt.subplots(nrows=nrows, ncols=ncols, figsize=(12, 8))
        flat = np.ravel(axs)
        for mode, per_mode in losses.items():
            for ax, (name, values) in zip(flat, per_mode.items()):
                ax.plot(values, lw=2, label=mode)
                ax.set_xlabel("Iterations")
                ax.set_title(name)
                ax.legend()
        fig.tight_layout()
    if show:
        plt.show()
    return fig, axs


def plot_images(images, labels=None, show=True, n=None):
    """Grid of up to 36 images (matplotlib; convenience only)."""
    import matplotlib.pyplot as plt
    n = images.shape[0] if n is None else n
    n = min(n, 36)
    nrows = max(int(np.sqrt(n)), 1)
    ncols = max(n // nrows, 1)
    fig, axs = plt.subplots(nrows=nrows, ncols=ncols, figsize=(10, 6))
    for i, ax in enumerate(np.ravel(axs)):
        ax.imshow(images[i])
        ax.axis("off")
        if labels is not None:
            ax.set_title("Label: {}".format(labels[i]))
    if show:
        plt.show()
    return fig, axs
