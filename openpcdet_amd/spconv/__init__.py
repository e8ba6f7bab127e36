"""MI355X-native stand-in for the `spconv` package namespace as pcdet imports it
(pcdet/utils/spconv_utils.py:3-10: `import spconv`, `float(spconv.__version__[2:]) >= 2.2`,
`spconv.constants.SPCONV_USE_DIRECT_TABLE = False`, `import spconv.pytorch as spconv`).

Use `openpcdet_amd.install_as_spconv()` to register it under the names `spconv`, `spconv.pytorch`, ... so the
reference's files import it unchanged; INTEGRATION.md shows the one-line alternative in spconv_utils.py.
"""
__version__ = "2.3.6"   # parsed as float("3.6") >= 2.2 by spconv_utils.py:4

from . import constants  # noqa: F401,E402
