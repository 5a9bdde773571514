"""reni_amd -- MI355X-native RENI forward/training hot path (HIP kernels behind the reference's
nn.Module surface).  See DESIGN.md."""
__version__ = "0.1.0"
