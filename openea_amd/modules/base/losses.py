"""Translational losses (mirror of openea/modules/base/losses.py:4-73).

The reference builds TF graph nodes; here each ``*_loss`` returns the ``oea_step_cfg`` fields
that select the same arithmetic inside the fused HIP step (csrc/triple_step.hip).
"""


def margin_loss(margin, loss_norm):
    """losses.py:15-27: sum relu(margin + s+ - s-)."""
    return dict(loss='margin-based', loss_norm=loss_norm, margin=margin)


def positive_loss(loss_norm):
    """losses.py:30-39: sum s+."""
    return dict(loss='positive', loss_norm=loss_norm)


def limited_loss(pos_margin, neg_margin, loss_norm, balance=1.0):
    """losses.py:42-56: sum relu(s+ - pos_margin) + balance * sum relu(neg_margin - s-)."""
    return dict(loss='limited', loss_norm=loss_norm, pos_margin=pos_margin, neg_margin=neg_margin, balance=balance)


def logistic_loss(loss_norm):
    """losses.py:59-73: sum log(1+exp(s+)) + sum log(1+exp(-s-))."""
    return dict(loss='logistic', loss_norm=loss_norm)


def alignment_loss():
    """approaches/bootea.py:197: -sum log sigmoid(-||h + r - t||^2)."""
    return dict(loss='align', loss_norm='L2')


def mapping_loss():
    """losses.py:76-80: sum ||e2 - e1 M||^2 + sum (M M^T - I)^2 (scaled by args.alpha in mapping.py:17) -- evaluated by
    oea_mapping_step."""
    return dict(loss='mapping', loss_norm='L2')


def get_loss_func(args):
    """losses.py:4-12 (note: balance is NOT passed on this path, losses.py:11 -> default 1.0)."""
    if args.loss == 'margin-based':
        return margin_loss(args.margin, args.loss_norm)
    if args.loss == 'logistic':
        return logistic_loss(args.loss_norm)
    if args.loss == 'limited':
        return limited_loss(args.pos_margin, args.neg_margin, args.loss_norm)
    return None
