"""Host-side engine: autograd operators over the HIP kernels, layer re-binding, fused optimizers, data parallel."""
