"""sibelia_amd -- MI355X-native BlockFinder hot path (de Bruijn graph build + bulge removal) of Sibelia.

Package contents: csrc/ (HIP kernels + the C ABI of include/sibelia_amd.h), api.py (host-side
mirror of the reference's BlockFinder interface over that ABI), workloads.py / formats.py
(synthetic inputs and canonical result serialisations used by tests and bench).
"""
from .api import BlockFinder, SibeliaError, load_library  # noqa: F401
