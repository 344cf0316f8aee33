"""Import shim: exposes the hyphen-named package directory ``smsut-medicalimgsegmentation_amd/`` as the
importable module ``smsut_amd`` (this file replaces itself in ``sys.modules`` with that package).

    import smsut_amd
    from smsut_amd.network.unet import UNet
    smsut_amd.install_dropin()      # optional: `from network.ugan import UGANnce` now resolves to this package
"""
import importlib
import importlib.util
import os
import sys

_PKG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "smsut-medicalimgsegmentation_amd")


def _load():
    spec = importlib.util.spec_from_file_location("smsut_amd", os.path.join(_PKG_DIR, "__init__.py"),
                                                  submodule_search_locations=[_PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["smsut_amd"] = mod
    spec.loader.exec_module(mod)
    mod.install_dropin = install_dropin
    mod.PACKAGE_DIR = _PKG_DIR
    return mod


_DROPIN = ("config", "network", "network.blocks", "network.unet", "network.ugan", "network.networks",
           "network.patchnce", "misc", "misc.loss", "misc.utils", "trainer", "trainer.baseTrainer",
           "trainer.unetTrainer", "trainer.uganShp0Trainer", "trainer.uganConsisTrainer", "trainer.uganTrainer",
           "trainer.meanTeacherTrainer", "trainer.crossPseTrainer")


def install_dropin():
    """Alias the reference's top-level module names (it runs with its repo root on sys.path:
    `from network.ugan import UGANnce`, `import config as cfg`, `from misc.loss import ...`) to this package."""
    for name in _DROPIN:
        sys.modules[name] = importlib.import_module("smsut_amd." + name)
    return sys.modules["network"]


_load()
