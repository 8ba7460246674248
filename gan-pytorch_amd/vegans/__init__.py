"""MI355X-native engine for the VeGANs adversarial training step (drop-in for `vegans` of tneuer/GAN-pytorch)."""
__version__ = "0.1.0"
