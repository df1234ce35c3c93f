"""sofima_amd: MI355X-native compute cores of SOFIMA (flow estimation + mesh relaxation).

`sofima_amd.flow_field` and `sofima_amd.mesh` mirror the public API of the
reference's `sofima.flow_field` and `sofima.mesh`; the kernels live in
libsofima_amd.so (HIP, gfx950) behind the C ABI in include/sofima_amd.h.
"""
from . import _abi  # noqa: F401
from . import flow_field  # noqa: F401
from . import flow_utils  # noqa: F401
from . import map_utils  # noqa: F401
from . import mesh  # noqa: F401
from . import stitch_elastic  # noqa: F401

__all__ = ['flow_field', 'flow_utils', 'map_utils', 'mesh', 'stitch_elastic']
