"""
baryonification_amd -- MI355X (gfx950) engine for BaryonForge's per-halo HEALPix-shell hot path.

Usage mirrors the reference package (`import BaryonForge as bfg`):

    import baryonification_amd as bfg
    Shell   = bfg.utils.LightconeShell(map=HealpixMap, cosmo=cosmo_dict)
    Catalog = bfg.utils.HaloLightConeCatalog(ra=ra, dec=dec, M=M200c, z=z, cosmo=cosmo_dict)
    new_map = bfg.Runners.BaryonifyShell(Catalog, Shell, epsilon_max=10, model=model).process()
"""
from . import _lib
from . import Profiles, Runners, utils
from .Profiles import *
from .Runners import *
from .utils import *

__version__ = '0.1.0'
