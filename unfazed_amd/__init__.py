"""unfazed_amd -- MI355X-native per-DNM phasing path (drop-in for unfazed's
informative_site_finder / read_collector / site_searcher / snv_phaser and the
allele-balance half of sv_phaser).

The compute path is hand-written HIP for gfx950 behind a C ABI
(include/unfazed_hip.h, unfazed_amd/csrc).  This package is the host side only:
decoded-input column builders, the ctypes binding and the mirrors of the
reference's phase_snvs / phase_svs / summarize_record surface.
"""
__version__ = "0.1.0"
# version of the reference whose surface is mirrored (reference unfazed/__init__.py:2)
__reference_version__ = "1.0.3"
