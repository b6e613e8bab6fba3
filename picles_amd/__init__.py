"""picles_amd — MI355X-native 2D particle-in-cell time step of the PiCLES wave model.

Host-side mirror of the reference's WaveGrowth2D / Simulation / Operators surface over the
C ABI of include/picles_hip.h (hand-written HIP kernels for gfx950).  See DESIGN.md.
"""
__version__ = "0.1.0"
