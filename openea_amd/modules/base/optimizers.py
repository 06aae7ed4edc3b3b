"""Optimiser selection (mirror of openea/modules/base/optimizers.py:4-20).

The reference builds `tf.train.{Adagrad,Adadelta,Adam,GradientDescent}Optimizer(learning_rate)` and calls
compute_gradients / apply_gradients on a loss.  Here the arithmetic lives in the kernels:
  * translational step (csrc/triple_step.hip): Adagrad (initial accumulator 0.1, no epsilon) and SGD fused on the touched
    rows; Adam and Adadelta as the dense pass TF runs for them (every row moves every step: the gather gradient comes
    back through l2_normalize as a dense tensor);
  * dense variables of the GNN approaches (alinet.py:871, rdgcn.py:332): `generate_optimizer(loss, lr, var_list, 'Adam')`
    returns the dense optimiser object over `var_list` (oea_adam_dense / oea_sgd_rows), whose step() plays
    apply_gradients.
"""
SUPPORTED = ('Adagrad', 'Adadelta', 'Adam', 'SGD')


def get_optimizer(opt, learning_rate):
    """optimizers.py:10-20: 'Adagrad' / 'Adadelta' / 'Adam', anything else -> SGD.  -> the fields of the step
    configuration that select the optimiser (ops.make_step_cfg(optimizer=..., lr=...))."""
    name = opt if opt in ('Adagrad', 'Adadelta', 'Adam') else 'SGD'
    return dict(optimizer=name, lr=learning_rate)


def generate_optimizer(loss, learning_rate, var_list=None, opt='SGD'):
    """optimizers.py:4-7.
    loss = a loss descriptor dict (modules/base/losses.py) -> the merged step configuration of the fused translational
    step; var_list = dense device parameters (torch tensors with .grad filled by the tape of models/graph_ops.py) -> an
    optimiser object over them."""
    spec = get_optimizer(opt, learning_rate)
    if var_list is not None and len(var_list) and hasattr(var_list[0], "is_cuda"):
        from ...models.graph_ops import DenseSGD, TFAdam
        if spec['optimizer'] == 'Adam':
            return TFAdam(var_list, learning_rate)
        if spec['optimizer'] == 'SGD':
            return DenseSGD(var_list, learning_rate)
        raise NotImplementedError("%s over dense variables: the GNN approaches use Adam (alinet.py:871, rdgcn.py:332) "
                                  "or SGD (gcn_align.py:511)" % spec['optimizer'])
    cfg = dict(loss or {})
    cfg.update(spec)
    return cfg
