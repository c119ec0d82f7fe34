"""tfep_amd: the tfep.nn normalizing-flow hot path on AMD Instinct MI355X (gfx950).

Drop-in for the reference's ``tfep.nn`` flows / ``tfep.loss`` / ``tfep.analysis.fep_estimator``
(same Module API and state_dict schema), computing in hand-written HIP kernels behind the C ABI
of ``include/tfep_hip.h``.  No CPU fallback: tensors must live on a HIP device.
"""
__version__ = '0.1.0'
