"""The plain translational ModelFamily members (openea/models/trans/__init__.py): TransE / TransH / TransD on the
fused device step.  (TransR -- a d x d matrix per relation, models/trans/transr.py -- is not built.)"""
from .transd import TransD  # noqa: F401
from .transe import TransE  # noqa: F401
from .transh import TransH  # noqa: F401
