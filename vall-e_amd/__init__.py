"""vall_e_amd -- MI355X-native AR+NAR forward of VALL-E behind the reference's API.

The directory is named ``vall-e_amd`` (repo convention); the importable module
name is ``vall_e_amd`` (see the one-file shim ``vall_e_amd.py`` at the repo root).
"""
__version__ = "0.1.0"
