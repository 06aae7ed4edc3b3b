"""Optimiser selection (mirror of openea/modules/base/optimizers.py:4-20).

Adagrad (initial accumulator 0.1, no epsilon) and SGD run inside the fused step kernel.
"""
SUPPORTED = ('Adagrad', 'SGD')


def get_optimizer(opt, learning_rate):
    """optimizers.py:10-20: anything that is not Adagrad / Adadelta / Adam falls back to SGD."""
    if opt in ('Adadelta', 'Adam'):
        raise NotImplementedError("%s is not used by the translational approaches on this path "
                                  "(MTransE / AlignE / BootEA use Adagrad)" % opt)
    name = 'Adagrad' if opt == 'Adagrad' else 'SGD'
    return dict(optimizer=name, lr=learning_rate)


def generate_optimizer(loss_cfg, learning_rate, var_list=None, opt='SGD'):
    """optimizers.py:4-7 -> merged step configuration dict."""
    cfg = dict(loss_cfg)
    cfg.update(get_optimizer(opt, learning_rate))
    return cfg
