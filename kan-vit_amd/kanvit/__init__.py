"""kanvit: Python binding of the MI355X hot-path library (see include/kanvit.h).

`ops` holds the autograd wrappers, `grouped` the helpers that launch one fused kernel for a
single KAN layer or for all heads' q|k|v mappings at once, `dp` the data-parallel gradient
reducer.  Importing this package does not load the shared library; the first op does, and
fails loudly if it is missing.
"""
from ._lib import KanvitError  # noqa: F401
