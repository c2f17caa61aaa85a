"""Moved into the package (bitnetmcu_amd/headerwriter.py); kept so that older scripts keep importing."""
from bitnetmcu_amd.headerwriter import write_header  # noqa: F401
