from .aligne import AlignE  # noqa: F401
from .alinet import AliNet  # noqa: F401
from .bootea import BootEA  # noqa: F401
from .bootea_rotate import BootEA_RotatE  # noqa: F401
from .bootea_transh import BootEA_TransH  # noqa: F401
from .gcn_align import GCN_Align  # noqa: F401
from .mtranse import MTransE  # noqa: F401
from .rdgcn import RDGCN  # noqa: F401
