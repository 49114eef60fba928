"""Import shim: `import ssunet_gan_amd` -> the package in the directory `ssunet-gan_amd/`
(a hyphen is not a legal identifier, so the directory cannot be imported by name)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'ssunet-gan_amd')
_spec = importlib.util.spec_from_file_location('ssunet_gan_amd', os.path.join(_dir, '__init__.py'),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules['ssunet_gan_amd'] = _mod
_spec.loader.exec_module(_mod)
