"""Importable alias of the `mpc-rl_for_avs_amd/` package directory.

The project directory carries the reference's name (with a hyphen, which Python cannot import);
this shim points the package search path at it so `import mpc_rl_for_avs_amd` works in place.
"""
import os as _os

__path__.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                                 "mpc-rl_for_avs_amd"))
