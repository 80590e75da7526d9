"""mimc3_amd -- MI355X-native DLC/NCC matcher + QM pseudo-smoothing for MIMC3 (hot path only).

Sub-modules:
  synth  -- seeded synthetic inputs (numpy/scipy only, no native code)
  api    -- host-side mirror of the reference entry points over the C-ABI library
            (csrc/libmimc3_hip.so).  Importing it without the built HIP library raises.
"""
__version__ = "0.1.0"
