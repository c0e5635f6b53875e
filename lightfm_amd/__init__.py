"""lightfm_amd -- MI355X-native (gfx950) engine behind LightFM's fit/predict API.

    from lightfm_amd import LightFM          # same constructor / fit / predict as lightfm.LightFM
    import lightfm_amd._lightfm_fast         # drop-in for lightfm._lightfm_fast (native module)
"""
from .lightfm import LightFM
from .options import options

__version__ = "0.1.0"
__all__ = ["LightFM", "options", "__version__"]
