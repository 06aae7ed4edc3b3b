"""openea_amd -- MI355X-native implementation of OpenEA's training / evaluation hot path.

Mirrors the reference's Python API for that path (``openea.modules.train.batch``,
``openea.modules.finding.*``, ``openea.models.basic_model.BasicModel`` and the translational /
GCN approaches) on top of libopenea_hip.so (include/openea_hip.h).  There is no CPU
fallback: compute entry points raise ``OpenEAHipError`` without the library or a GPU.
"""
from ._lib import OpenEAHipError  # noqa: F401

__version__ = "0.1.0"
