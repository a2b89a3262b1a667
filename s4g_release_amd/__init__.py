"""s4g_release_amd -- MI355X-native PointNet++ SA/FP hot path of S4G.

(The task names the package ``s4g-release_amd``; a hyphen cannot be imported,
so the importable spelling uses an underscore.)

Importing the package does not touch the GPU and does not load the HIP library;
the first operator call does, and fails loudly if ``libs4g_hip.so`` is absent.
"""
__version__ = "0.1.0"
