"""MI355X-native implementation of the SMSUT conv hot path (U-Net segmentor + ugan translation GAN).

The directory name carries a hyphen, so import it through ``smsut_amd.py`` at the repository root
(``import smsut_amd``), or call ``smsut_amd.install_dropin()`` to make the reference's own import
statements (``from network.ugan import UGANnce`` ...) resolve to this package.
"""
from . import _hip  # noqa: F401
