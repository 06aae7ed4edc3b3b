"""Mapping module of MTransE under the reference's names (mirror of openea/modules/base/mapping.py:9-25): the variables
(orthogonal d x d matrix + identity) and the graph (alpha * (sum ||e2 - e1 M||^2 + ||M M^T - I||^2), its own optimiser)
are the model's `_define_mapping_variables` / `_define_mapping_graph`; the step itself is oea_mapping_step."""


def add_mapping_variables(model):
    """mapping.py:22-25."""
    model._define_mapping_variables()


def add_mapping_module(model):
    """mapping.py:9-19."""
    model._define_mapping_graph()
