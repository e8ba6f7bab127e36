"""spconv.constants: pcdet/utils/spconv_utils.py:4-5 assigns SPCONV_USE_DIRECT_TABLE = False; accepted, no effect
(output rows are always in ascending key order here)."""
SPCONV_USE_DIRECT_TABLE = False
